// Developer micro-benchmark: calibrates the rocprofv3 FETCH_SIZE counter on gfx950 for the access shapes this library uses
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads; other widths are
// uncalibrated").  Every kernel reads a known number of bytes ONCE from a buffer far larger than the Infinity Cache (1 GiB), so
// FETCH_SIZE x factor = bytes tells the factor per access shape:
//   read16 / read8 / read4 / read2: fully coalesced streaming reads of 16 / 8 / 4 / 2 bytes per lane
//   seg288: 288-byte row segments at a 1280-byte pitch read with 16-byte loads (the MFMA stem's patch rows of a 640-px f16 image)
// run:  rocprofv3 --pmc FETCH_SIZE -d out -o p --output-format csv -- ./fetch_calib     (tools/fetch_calib.sh prints the table)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <typename V> __global__ __launch_bounds__(256) void read_k(const V* __restrict__ a, long n, unsigned* sink) {
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const V v = a[i];
    const unsigned char* p = reinterpret_cast<const unsigned char*>(&v);
    acc += p[0];
  }
  if (acc == 0xFFFFFFFFu) *sink = acc;
}
// 288-byte segments (18 x 16 B) at a 1280-byte pitch: thread -> (row, 16-byte piece)
__global__ __launch_bounds__(256) void seg288(const uint4* __restrict__ a, long rows, unsigned* sink) {
  unsigned acc = 0;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < rows * 18; t += (long)gridDim.x * 256) {
    const long row = t / 18, piece = t - row * 18;
    const uint4 v = a[row * 80 + piece];
    acc += v.x & 1;
  }
  if (acc == 0xFFFFFFFFu) *sink = acc;
}
int main() {
  const long bytes = 1L << 30;
  void* buf; unsigned* sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 1, bytes));
  CK(hipDeviceSynchronize());
  const int grid = 256 * 8;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(read_k<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(read_k<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, sink);
    hipLaunchKernelGGL(read_k<unsigned>, dim3(grid), dim3(256), 0, 0, (const unsigned*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(read_k<unsigned short>, dim3(grid), dim3(256), 0, 0, (const unsigned short*)buf, bytes / 2, sink);
    hipLaunchKernelGGL(seg288, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 1280, sink);
  }
  CK(hipDeviceSynchronize());
  printf("bytes read per kernel: read16/8/4/2 = %ld, seg288 = %ld useful of %ld spanned\n", bytes, bytes / 1280 * 288, bytes);
  return 0;
}
