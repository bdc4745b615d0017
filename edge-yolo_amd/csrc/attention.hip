// K8a linear attention core and K8b softmax attention core (fp32 arithmetic, wave64 reductions).
//
// Linear attention (LinearAttention.forward, block.py:3360-3373), per (image, head), head_dim d <= 64:
//   ks[n][i] = softmax_i(k[n][:])            (over head_dim, one pixel = one wave: 64 lanes <-> 64 channels)
//   ctx[i][j] = sum_n ks[n][i] * v[n][j]     (d x d, fp32, lane j keeps column j in registers)
//   qs[n][i] = softmax_n(q[:][i])            (over ALL pixels: two-pass column reduction, max/sum then normalise)
//   y[n][j]  = sum_i qs[n][i] * ctx[i][j]
// One workgroup (8 waves) per (image, head); N is streamed, so N=1600 (1280^2 input) needs no more LDS than N=400.
#include "common.h"

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// LA_WAVES waves per (image, head); a wave takes every LA_WAVES-th pixel.  Cross-lane traffic goes through per-wave LDS rows
// (one broadcast ds_read_b128 feeds 4 FMAs) instead of 64 ds_bpermute per pixel.
#define LA_WAVES 8
#define LA_U 4
template <typename T>
__global__ __launch_bounds__(64 * LA_WAVES) void linattn_kernel(int N, int C, int heads, int d, const T* __restrict__ qkv, int qCs, T* __restrict__ y, int yCs) {
  __shared__ float ctx[64][64];
  __shared__ float red[LA_WAVES][64];
  __shared__ __attribute__((aligned(16))) float rowbuf[LA_WAVES][64];
  __shared__ float qmax[64], qsum[64];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool act = lane < d;
  const T* base = qkv + (long)b * N * qCs + h * d + lane;
  const T* qp = base;
  const T* kp = base + C;
  const T* vp = base + 2 * C;

  // ---- q column statistics over N (max, then sum of exp)
  // (pixel loops are unrolled by LA_U so that LA_U independent loads are in flight: each iteration is one dependent
  //  round trip to L2/HBM otherwise)
  float m = -INFINITY;
  for (int n0 = wave; n0 < N; n0 += LA_WAVES * LA_U) {
    float t[LA_U];
#pragma unroll
    for (int u = 0; u < LA_U; ++u) { const int n = n0 + u * LA_WAVES; t[u] = (act && n < N) ? to_f(qp[(long)n * qCs]) : -INFINITY; }
#pragma unroll
    for (int u = 0; u < LA_U; ++u) m = fmaxf(m, t[u]);
  }
  red[wave][lane] = m;
  for (int i = threadIdx.x; i < 64 * 64; i += 64 * LA_WAVES) ctx[i >> 6][i & 63] = 0.f;
  __syncthreads();
  if (wave == 0) {
    float mm = red[0][lane];
#pragma unroll
    for (int w = 1; w < LA_WAVES; ++w) mm = fmaxf(mm, red[w][lane]);
    qmax[lane] = mm;
  }
  __syncthreads();
  const float qm = qmax[lane];
  float s = 0.f;
  for (int n0 = wave; n0 < N; n0 += LA_WAVES * LA_U) {
    float t[LA_U];
#pragma unroll
    for (int u = 0; u < LA_U; ++u) { const int n = n0 + u * LA_WAVES; t[u] = (act && n < N) ? to_f(qp[(long)n * qCs]) : -INFINITY; }
#pragma unroll
    for (int u = 0; u < LA_U; ++u) s += __expf(t[u] - qm);  // exp(-inf) = 0 for the masked ones
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0) {
    float ss = 0.f;
#pragma unroll
    for (int w = 0; w < LA_WAVES; ++w) ss += red[w][lane];
    qsum[lane] = ss;
  }

  // ---- ctx partial per wave: lane j accumulates column j over this wave's pixels
  float col[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) col[i] = 0.f;
  float* rb = rowbuf[wave];
  for (int n0 = wave; n0 < N; n0 += LA_WAVES * LA_U) {
    float kk[LA_U], vu[LA_U];
#pragma unroll
    for (int u = 0; u < LA_U; ++u) {
      const int n = n0 + u * LA_WAVES;
      const bool ok = act && n < N;
      kk[u] = ok ? to_f(kp[(long)n * qCs]) : -INFINITY;
      vu[u] = ok ? to_f(vp[(long)n * qCs]) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < LA_U; ++u) {
      if (n0 + u * LA_WAVES >= N) break;  // wave-uniform
      const float km = wave_max(kk[u]);
      const float e = act ? __expf(kk[u] - km) : 0.f;
      rb[lane] = e * __builtin_amdgcn_rcpf(wave_sum(e));
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i4 = 0; i4 < 16; ++i4) {
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(rb + 4 * i4);  // same address in every lane: LDS broadcast
#pragma unroll
        for (int j = 0; j < 4; ++j) col[4 * i4 + j] += k4[j] * vu[u];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // reduce the per-wave partial ctx in a FIXED wave order (deterministic sums; ctx was zeroed above)
  for (int w = 0; w < LA_WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 64; ++i) ctx[i][lane] += col[i];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 64; ++i) col[i] = ctx[i][lane];
  const float qinv = __builtin_amdgcn_rcpf(qsum[lane]);

  // ---- y = softmax_N(q) @ ctx
  T* yp = y + (long)b * N * yCs + h * d + lane;
  for (int n0 = wave; n0 < N; n0 += LA_WAVES * LA_U) {
    float qq[LA_U];
#pragma unroll
    for (int u = 0; u < LA_U; ++u) { const int n = n0 + u * LA_WAVES; qq[u] = (act && n < N) ? to_f(qp[(long)n * qCs]) : -INFINITY; }
#pragma unroll
    for (int u = 0; u < LA_U; ++u) {
      const int n = n0 + u * LA_WAVES;
      if (n >= N) break;  // wave-uniform
      rb[lane] = __expf(qq[u] - qm) * qinv;
      __builtin_amdgcn_wave_barrier();
      float o = 0.f;
#pragma unroll
      for (int i4 = 0; i4 < 16; ++i4) {
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(rb + 4 * i4);
#pragma unroll
        for (int j = 0; j < 4; ++j) o += q4[j] * col[4 * i4 + j];
      }
      __builtin_amdgcn_wave_barrier();
      if (act) yp[(long)n * yCs] = from_f<T>(o);
    }
  }
}

extern "C" int ey_linear_attention(int dtype, int B, int N, int C, int heads, const void* qkv, int qkv_cstride, void* y, int y_cstride, ey_stream_t stream) {
  EY_CHECK(qkv && y, "linear_attention: null pointer");
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "linear_attention: bad dtype");
  EY_CHECK(B > 0 && N > 0 && heads > 0 && C % heads == 0, "linear_attention: B=%d N=%d C=%d heads=%d", B, N, C, heads);
  const int d = C / heads;
  if (d > 64) return ey_set_error(EY_EUNSUPPORTED, "linear_attention: head_dim %d > 64", d);
  EY_CHECK(qkv_cstride >= 3 * C && y_cstride >= C, "linear_attention: cstride");
  dim3 grid(B * heads);
  if (dtype == EY_F16) hipLaunchKernelGGL(linattn_kernel<f16>, grid, dim3(64 * LA_WAVES), 0, (hipStream_t)stream, N, C, heads, d, (const f16*)qkv, qkv_cstride, (f16*)y, y_cstride);
  else hipLaunchKernelGGL(linattn_kernel<float>, grid, dim3(64 * LA_WAVES), 0, (hipStream_t)stream, N, C, heads, d, (const float*)qkv, qkv_cstride, (float*)y, y_cstride);
  EY_LAUNCH_CHECK("ey_linear_attention");
  return EY_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Softmax attention core (Attention.forward, block.py:1042-1053).  Per (image, head): channels of that head are
// [q (kd) | k (kd) | v (hd)].  For key/column index m:  attn[n][m] = softmax_m(scale * q[:,n].k[:,m]);
// y[c][n] = sum_m v[c][m] * attn[n][m].  One wave handles one query pixel n at a time: lanes stride over m for the
// scores (online max/sum not needed: two passes over LDS-resident scores), then over output channels c.
// K and V of the (image, head) are staged once in LDS as fp32 when they fit; N is tiled otherwise.
template <typename T, bool K_LDS>
__global__ __launch_bounds__(256) void softattn_kernel(int N, int heads, int kd, int hd, float scale, const T* __restrict__ qkv, int qCs,
                                                       T* __restrict__ y, int yCs, int n_per_block) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int per = 2 * kd + hd;
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* Sc = reinterpret_cast<float*>(smem);  // [4][N] scores of the wave's current query
  float* Qs = Sc + 4L * N;                      // [4][64]
  T* Ks = reinterpret_cast<T*>(Qs + 4 * 64);    // [N][kd] (only when K_LDS)
  const T* base = qkv + (long)b * N * qCs + h * per;
  if (K_LDS) {
    for (int i = threadIdx.x; i < N * kd; i += blockDim.x) {
      const int n = i / kd, c = i - n * kd;
      Ks[i] = base[(long)n * qCs + kd + c];
    }
    __syncthreads();
  }
  float* sc = Sc + (long)wave * N;
  float* qs = Qs + wave * 64;
  const int n0 = blockIdx.y * n_per_block, n1 = min(N, n0 + n_per_block);
  for (int n = n0 + wave; n < n1; n += 4) {
    // (no cross-lane shuffles inside the m-loops: they are divergent when N is not a multiple of 64)
    qs[lane] = lane < kd ? to_f(base[(long)n * qCs + lane]) * scale : 0.f;
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int m = lane; m < N; m += 64) {
      float a = 0.f;
      if (K_LDS) {
        for (int c = 0; c < kd; ++c) a += qs[c] * to_f(Ks[m * kd + c]);
      } else {
        const T* kp = base + (long)m * qCs + kd;
        for (int c = 0; c < kd; ++c) a += qs[c] * to_f(kp[c]);
      }
      sc[m] = a;
      mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int m = lane; m < N; m += 64) {
      const float e = __expf(sc[m] - mx);
      sc[m] = e;
      sum += e;
    }
    const float inv = 1.f / wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    // sc[] is written and read by this wave only.  V rows are read from global (L1/L2-resident: N*hd elements)
    for (int c = lane; c < hd; c += 64) {
      const T* vp = base + 2 * kd + c;
      float o = 0.f;
      for (int m = 0; m < N; ++m) o += sc[m] * to_f(vp[(long)m * qCs]);
      y[((long)b * N + n) * yCs + h * hd + c] = from_f<T>(o * inv);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" int ey_softmax_attention(int dtype, int B, int N, int heads, int kd, int hd, float scale, const void* qkv, int qkv_cstride, void* y, int y_cstride,
                                    ey_stream_t stream) {
  EY_CHECK(qkv && y, "softmax_attention: null pointer");
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "softmax_attention: bad dtype");
  EY_CHECK(B > 0 && N > 0 && heads > 0 && kd > 0 && hd > 0, "softmax_attention: bad extent");
  if (kd > 64) return ey_set_error(EY_EUNSUPPORTED, "softmax_attention: key_dim %d > 64", kd);
  EY_CHECK(qkv_cstride >= heads * (2 * kd + hd) && y_cstride >= heads * hd, "softmax_attention: cstride");
  const int es = dtype == EY_F16 ? 2 : 4;
  const size_t lds_base = (4 * (size_t)N + 4 * 64) * 4;
  if (lds_base > 160 * 1024) return ey_set_error(EY_EUNSUPPORTED, "softmax_attention: N=%d needs %zu B of LDS", N, lds_base);
  const bool k_lds = lds_base + (size_t)N * kd * es <= 160 * 1024;
  const size_t lds = lds_base + (k_lds ? (size_t)N * kd * es : 0);
  const int nsplit = B * heads >= 512 ? 1 : (512 + B * heads - 1) / (B * heads);
  const int n_per_block = (N + nsplit - 1) / nsplit;
  dim3 grid(B * heads, (N + n_per_block - 1) / n_per_block);
  hipStream_t st = (hipStream_t)stream;
#define SOFTATT(T, KL)                                                                                                                   \
  do {                                                                                                                                   \
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)softattn_kernel<T, KL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return ey_set_error(EY_ELAUNCH, "cannot reserve %zu B of LDS", lds);                                                               \
    hipLaunchKernelGGL((softattn_kernel<T, KL>), grid, dim3(256), lds, st, N, heads, kd, hd, scale, (const T*)qkv, qkv_cstride, (T*)y, y_cstride, n_per_block); \
  } while (0)
  if (dtype == EY_F16) { if (k_lds) SOFTATT(f16, true); else SOFTATT(f16, false); }
  else { if (k_lds) SOFTATT(float, true); else SOFTATT(float, false); }
#undef SOFTATT
  EY_LAUNCH_CHECK("ey_softmax_attention");
  return EY_OK;
}
