// K8a linear attention core and K8b softmax attention core (fp32 arithmetic, wave64 reductions).
//
// Linear attention (LinearAttention.forward, block.py:3360-3373), per (image, head), head_dim d <= 64:
//   ks[n][i] = softmax_i(k[n][:])            (over head_dim, one pixel = one wave: 64 lanes <-> 64 channels)
//   ctx[i][j] = sum_n ks[n][i] * v[n][j]     (d x d, fp32, lane j keeps column j in registers)
//   qs[n][i] = softmax_n(q[:][i])            (over ALL pixels: two-pass column reduction, max/sum then normalise)
//   y[n][j]  = sum_i qs[n][i] * ctx[i][j]
// One workgroup (8 waves) per (image, head); N is streamed, so N=1600 (1280^2 input) needs no more LDS than N=400.
#include "common.h"
#include "tune.h"
#include <stdlib.h>

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

// ---------------------------------------------------------------------------------------------------------------
// MFMA form of the linear attention core (f16 mode, head_dim 64).  The VALU kernel above spends its time on 2 x 64 FMAs per
// pixel per lane plus wave-wide softmax reductions on 64 workgroups; both products are small GEMMs:
//   ctx[i][j] = sum_n ks[n][i] v[n][j]   (contraction over pixels)        y[n][j] = sum_i qs[n][i] ctx[i][j]
// One 512-thread workgroup per (image, head):
//   A. q column statistics (max, sum exp over all pixels): lane = (pixel of 8, channel octet), 16-byte loads, shuffle + LDS reduce
//   B. per 128-pixel chunk: k softmax over the 64 channels (8 lanes per pixel, 3 shuffle steps), ks (f16) and v -> LDS [n][72]
//   C. ctx += ks^T v on MFMA: wave w owns output blocks (i-block w&3, j-blocks 2(w>>2), 2(w>>2)+1) for ALL pixels, so no
//      cross-wave reduction exists; the operands need the pixel index register-contiguous -> 8 ds_read_u16 per fragment
//   D. ctx (fp32 accumulators) -> LDS as f16 ctxT[j][i]
//   E. y = qs ctx on MFMA: B fragment = exp(q - qmax) / qsum computed in registers from 16-byte q loads (k = channel, contiguous),
//      A fragments = ctxT rows (register resident), result lane = pixel -> 8-byte stores.
#include "linattn_mfma.inc.h"
__global__ __launch_bounds__(512) void linattn_mfma_kernel(int N, int C, int heads, const f16* __restrict__ qkv, int qCs, f16* __restrict__ y, int yCs) {
  __shared__ LinAttnLds S;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const f16* qb = qkv + (long)b * N * qCs + h * 64;
  linattn_mfma_head(S, N, qb, qb + C, qb + 2 * C, qCs, y + (long)b * N * yCs + h * 64, yCs, threadIdx.x);
}

extern "C" int ey_linear_attention(int dtype, int B, int N, int C, int heads, const void* qkv, int qkv_cstride, void* y, int y_cstride, ey_stream_t stream) {
  EY_CHECK(qkv && y, "linear_attention: null pointer");
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "linear_attention: bad dtype");
  EY_CHECK(B > 0 && N > 0 && heads > 0 && C % heads == 0, "linear_attention: B=%d N=%d C=%d heads=%d", B, N, C, heads);
  const int d = C / heads;
  if (d > 64) return ey_set_error(EY_EUNSUPPORTED, "linear_attention: head_dim %d > 64", d);
  EY_CHECK(qkv_cstride >= 3 * C && y_cstride >= C, "linear_attention: cstride");
  dim3 grid(B * heads);
  const bool mfma_off = !tune().linattn_mfma;
  if (dtype == EY_F16 && d == 64 && !mfma_off && (qkv_cstride * 2) % 16 == 0 && ey_aligned(qkv, 16) && C % 8 == 0 && (y_cstride * 2) % 8 == 0 && ey_aligned(y, 8)) {
    hipLaunchKernelGGL(linattn_mfma_kernel, grid, dim3(512), 0, (hipStream_t)stream, N, C, heads, (const f16*)qkv, qkv_cstride, (f16*)y, y_cstride);
    EY_LAUNCH_CHECK("ey_linear_attention(mfma)");
    return EY_OK;
  }
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

// ---------------------------------------------------------------------------------------------------------------
// MFMA form of the softmax attention core (f16 mode, key_dim 32, head_dim 64, N <= 416: the C2PSA block of the YOLO11 baseline
// at 640x640).  The VALU kernel above was 35 % of the YOLO11n step (0.85 ms).  One 512-thread workgroup per (image, head); K and V
// of the head are staged once in LDS; a wave owns 16 queries at a time, flash-attention style but with all scores in registers:
//   S^T = K Q^T            A = K rows (keys) from LDS, B = Q (16-byte loads, k = the 32 key channels): lane = query, 4 keys per block
//   softmax over keys      per-lane over its 4*NKB values, then across the 4 lanes that share a query (2 shuffles)
//   Y^T = V^T P^T          contraction over keys in the ORDER the scores already sit in the registers (k-slot 8g+j of a 32-key step
//                          = key 32s + 16(j>>2) + 4g + (j&3)), so P needs no cross-lane movement; V^T fragments by 2-byte LDS gathers
#define SM_MAXKS 13        // 32-key steps: N <= 416
#define SM_KLS 40          // LDS row stride of K (elements)
#define SM_VLS 72          // LDS row stride of V
template <int NKS>
__global__ __launch_bounds__(512) void softattn_mfma_kernel(int N, int heads, float scale, const f16* __restrict__ qkv, int qCs, f16* __restrict__ y, int yCs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* Ks = reinterpret_cast<f16*>(smem);      // [NKS*32][SM_KLS]
  f16* Vs = Ks + NKS * 32 * SM_KLS;            // [NKS*32][SM_VLS]
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const f16* base = qkv + (long)b * N * qCs + h * 128;  // [q 32 | k 32 | v 64]
  for (int v = tid; v < NKS * 32 * 4; v += 512) {
    const int n = v >> 2, cv = v & 3;
    Vec8<f16> t;
    if (n < N) t.load(base + (long)n * qCs + 32 + cv * 8);
    else t.zero();
    t.store(Ks + n * SM_KLS + cv * 8);
  }
  for (int v = tid; v < NKS * 32 * 8; v += 512) {
    const int n = v >> 3, cv = v & 7;
    Vec8<f16> t;
    if (n < N) t.load(base + (long)n * qCs + 64 + cv * 8);
    else t.zero();
    t.store(Vs + n * SM_VLS + cv * 8);
  }
  __syncthreads();
  for (int qb = wave; qb * 16 < N; qb += 8) {
    const int n = qb * 16 + r;
    Vec8<f16> qf;
    if (n < N) qf.load(base + (long)n * qCs + 8 * g);
    else qf.zero();
    // scores: S[kb][t] = q(n) . k(16 kb + 4g + t)
    f32x4 S[2 * NKS];
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2 * NKS; ++kb) {
      Vec8<f16> kf;
      kf.load(Ks + (kb * 16 + r) * SM_KLS + 8 * g);
      S[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf.v, qf.v, (f32x4)0.f, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (kb * 16 + 4 * g + t >= N) S[kb][t] = -INFINITY;  // padded keys
        mx = fmaxf(mx, S[kb][t]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2 * NKS; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) { S[kb][t] = __expf((S[kb][t] - mx) * scale); sum += S[kb][t]; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = __builtin_amdgcn_rcpf(sum);
    // y^T[c][n] = sum_key v[key][c] p[n][key]
    f32x4 acc[4] = {(f32x4)0.f, (f32x4)0.f, (f32x4)0.f, (f32x4)0.f};
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      Vec8<f16> pf;
#pragma unroll
      for (int t = 0; t < 4; ++t) { pf.v[t] = (f16)(S[2 * s][t] * inv); pf.v[4 + t] = (f16)(S[2 * s + 1][t] * inv); }
      const f16* vp = Vs + (32 * s + 4 * g) * SM_VLS + r;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        Vec8<f16> vf;
#pragma unroll
        for (int j = 0; j < 8; ++j) vf.v[j] = vp[(16 * (j >> 2) + (j & 3)) * SM_VLS + cb * 16];
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf.v, pf.v, acc[cb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the gathers of later steps from being hoisted over the 100+ live score registers
    }
    if (n < N) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const f16x4 o = {(f16)acc[cb][0], (f16)acc[cb][1], (f16)acc[cb][2], (f16)acc[cb][3]};
        *reinterpret_cast<f16x4*>(y + ((long)b * N + n) * yCs + h * 64 + cb * 16 + 4 * g) = o;
      }
    }
  }
}

template <int NKS>
static int softattn_mfma_launch(int B, int N, int heads, float scale, const void* qkv, int qCs, void* y, int yCs, hipStream_t st) {
  const size_t lds = (size_t)NKS * 32 * (SM_KLS + SM_VLS) * sizeof(f16);
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)softattn_mfma_kernel<NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    attr = true;
  }
  hipLaunchKernelGGL((softattn_mfma_kernel<NKS>), dim3(B * heads), dim3(512), lds, st, N, heads, scale, (const f16*)qkv, qCs, (f16*)y, yCs);
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_softmax_attention(mfma): %s", hipGetErrorString(e_));
  return 1;
}

extern "C" int ey_softmax_attention(int dtype, int B, int N, int heads, int kd, int hd, float scale, const void* qkv, int qkv_cstride, void* y, int y_cstride,
                                    ey_stream_t stream) {
  EY_CHECK(qkv && y, "softmax_attention: null pointer");
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "softmax_attention: bad dtype");
  EY_CHECK(B > 0 && N > 0 && heads > 0 && kd > 0 && hd > 0, "softmax_attention: bad extent");
  if (kd > 64) return ey_set_error(EY_EUNSUPPORTED, "softmax_attention: key_dim %d > 64", kd);
  EY_CHECK(qkv_cstride >= heads * (2 * kd + hd) && y_cstride >= heads * hd, "softmax_attention: cstride");
  const int es = dtype == EY_F16 ? 2 : 4;
  const bool sm_mfma_off = !tune().softattn_mfma;
  if (dtype == EY_F16 && kd == 32 && hd == 64 && N <= 32 * SM_MAXKS && !sm_mfma_off && (qkv_cstride * 2) % 16 == 0 && ey_aligned(qkv, 16) && (y_cstride * 2) % 8 == 0 &&
      ey_aligned(y, 8)) {
    const int nks = (N + 31) / 32;
    int rc = 0;
    // the score registers are sized at compile time: 4 variants cover N <= 128 / 256 / 320 / 416
    if (nks <= 4) rc = softattn_mfma_launch<4>(B, N, heads, scale, qkv, qkv_cstride, y, y_cstride, (hipStream_t)stream);
    else if (nks <= 8) rc = softattn_mfma_launch<8>(B, N, heads, scale, qkv, qkv_cstride, y, y_cstride, (hipStream_t)stream);
    else if (nks <= 10) rc = softattn_mfma_launch<10>(B, N, heads, scale, qkv, qkv_cstride, y, y_cstride, (hipStream_t)stream);
    else rc = softattn_mfma_launch<13>(B, N, heads, scale, qkv, qkv_cstride, y, y_cstride, (hipStream_t)stream);
    if (rc != 0) return rc < 0 ? rc : EY_OK;
  }
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
