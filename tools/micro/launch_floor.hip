// Developer micro-benchmark: per-launch floor of dependent small kernels replayed from a hipGraph on one stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_empty() {}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i];
}
// one wave per 16 "pixels": dependent chain load -> load -> store (two round trips)
__global__ __launch_bounds__(256) void k_chain(const uint4* __restrict__ a, uint4* __restrict__ b, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { uint4 v = a[i]; uint4 w = a[(i + (v.x & 1)) % n]; v.x += w.y; b[i] = v; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <typename F> float run_graph(hipStream_t st, int nk, F launch, int reps = 20) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < nk; ++i) launch(i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3f / (reps * nk);
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int NB = 8; const size_t bytes = 64 << 20;
  std::vector<uint4*> buf(NB);
  for (auto& p : buf) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); }
  const int nk = 120;
  printf("empty kernel, 1 block: %.2f us/launch\n", run_graph(st, nk, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }));
  printf("empty kernel, 2048 blocks x256: %.2f us/launch\n", run_graph(st, nk, [&](int) { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, st); }));
  for (long kb : {256L, 1024L, 4096L, 16384L, 65536L}) {
    long n = kb * 1024 / 16;
    float c = run_graph(st, nk, [&](int i) { hipLaunchKernelGGL(k_copy, dim3((n + 255) / 256), dim3(256), 0, st, buf[i % NB], buf[(i + 1) % NB], n); });
    float d = run_graph(st, nk, [&](int i) { hipLaunchKernelGGL(k_chain, dim3((n + 255) / 256), dim3(256), 0, st, buf[i % NB], buf[(i + 1) % NB], n); });
    printf("copy %6ld KB: %.2f us/launch (%.0f GB/s)   chain(2 trips): %.2f us/launch\n", kb, c, 2.0 * kb * 1024 / c / 1e3, d);
  }
  return 0;
}
