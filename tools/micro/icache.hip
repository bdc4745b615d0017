// Developer micro-benchmark: cost of cold instruction fetch for short kernels with long straight-line code.
// N distinct kernels (template ID), each executing ~INSTR unrolled dependent-free FMAs once per wave, replayed from a graph
// (a) the same kernel back to back (warm i-cache)  (b) cycling through 16 distinct kernels of the same size (cold).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ID, int INSTR>
__global__ __launch_bounds__(256) void k_code(float* out, float s) {
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = s + j + ID;
#pragma unroll
  for (int i = 0; i < INSTR / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = __builtin_fmaf(a[j], s, (float)(i * 8 + j + ID * 7919));
  }
  float r = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) r += a[j];
  if (r == 12345.678f) out[threadIdx.x] = r;
}
typedef void (*kfn)(float*, float);
template <int INSTR> struct Tab {
  static kfn get(int id) {
    switch (id & 15) {
#define C(i) case i: return k_code<i, INSTR>;
      C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
    }
    return nullptr;
  }
};
template <typename F> float run_graph(hipStream_t st, int nk, F launch, int reps = 20) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < nk; ++i) launch(i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, st);
  (void)hipStreamSynchronize(st);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) (void)hipGraphLaunch(ge, st);
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / (reps * nk);
}
template <int INSTR> void test(hipStream_t st, float* out, int blocks) {
  float warm = run_graph(st, 128, [&](int) { hipLaunchKernelGGL(Tab<INSTR>::get(0), dim3(blocks), dim3(256), 0, st, out, 1.0001f); });
  float cold = run_graph(st, 128, [&](int i) { hipLaunchKernelGGL(Tab<INSTR>::get(i), dim3(blocks), dim3(256), 0, st, out, 1.0001f); });
  printf("code ~%5d B (%5d fma), %4d blocks: same kernel %.2f us/launch, 16 distinct kernels cycled %.2f us/launch\n", INSTR * 8, INSTR, blocks, warm, cold);
}
int main() {
  hipStream_t st; (void)hipStreamCreate(&st);
  float* out; (void)hipMalloc(&out, 4096);
  for (int blocks : {256, 1024}) {
    test<256>(st, out, blocks);
    test<1024>(st, out, blocks);
    test<2048>(st, out, blocks);
    test<4096>(st, out, blocks);
  }
  return 0;
}
