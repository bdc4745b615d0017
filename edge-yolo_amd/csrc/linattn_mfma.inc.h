// MFMA linear-attention core shared by attention.hip (one workgroup per (image, head)) and block.hip (per-image block programs).
#pragma once
#include "common.h"
#define LM_CH 128  // pixels per chunk
#define LM_LS 72   // LDS row stride of ks / vs / ctxT (elements)
__device__ __forceinline__ float oct_max(float v) {  // over the 8 lanes that share a pixel (lane bits 0..2)
  v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); return fmaxf(v, __shfl_xor(v, 4));
}
__device__ __forceinline__ float oct_sum(float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v + __shfl_xor(v, 4); }

// LDS of one (image, head) problem; a 1024-thread block kernel (block.hip) runs two heads side by side with two of these.
struct LinAttnLds {
  __attribute__((aligned(16))) f16 ks[LM_CH * LM_LS];
  __attribute__((aligned(16))) f16 vs[LM_CH * LM_LS];
  __attribute__((aligned(16))) f16 ctxT[64 * LM_LS];
  float red[8][64];
  float qmax[64], qinv[64];
};
// One head of one image by 512 threads (tid 0..511, 8 waves).  qb/kb/vb: this head's q / k / v channel 0 of pixel 0; yb: this head's
// output channel 0 of pixel 0.  Every thread of the workgroup must call it with the same N (it synchronises with __syncthreads()).
__device__ __forceinline__ void linattn_mfma_head(LinAttnLds& S, int N, const f16* __restrict__ qb, const f16* __restrict__ kb, const f16* __restrict__ vb, int qCs,
                                                  f16* __restrict__ yb, int yCs, int tid) {
  f16* ks = S.ks; f16* vs = S.vs; f16* ctxT = S.ctxT;
  float (*red)[64] = S.red;
  float* qmax = S.qmax; float* qinv = S.qinv;
  const int lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int pr = lane >> 3, co = lane & 7;  // elementwise phases: pixel-in-group, channel octet
  // ---- A. q column max / sum(exp)
  {
    float m8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m8[i] = -INFINITY;
    for (int n = wave * 8 + pr; n < N; n += 64) {
      Vec8<f16> q;
      q.load(qb + (long)n * qCs + co * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) m8[i] = fmaxf(m8[i], q.get(i));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // over the 8 pixel lanes (lane bits 3..5)
      m8[i] = fmaxf(m8[i], __shfl_xor(m8[i], 8)); m8[i] = fmaxf(m8[i], __shfl_xor(m8[i], 16)); m8[i] = fmaxf(m8[i], __shfl_xor(m8[i], 32));
    }
    if (pr == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave][co * 8 + i] = m8[i];
    }
    __syncthreads();
    if (tid < 64) {
      float m = red[0][tid];
#pragma unroll
      for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w][tid]);
      qmax[tid] = m;
    }
    __syncthreads();
    float qm[8], s8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { qm[i] = qmax[co * 8 + i]; s8[i] = 0.f; }
    for (int n = wave * 8 + pr; n < N; n += 64) {
      Vec8<f16> q;
      q.load(qb + (long)n * qCs + co * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) s8[i] += __expf(q.get(i) - qm[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { s8[i] += __shfl_xor(s8[i], 8); s8[i] += __shfl_xor(s8[i], 16); s8[i] += __shfl_xor(s8[i], 32); }
    __syncthreads();  // red[] reads of the max pass are done
    if (pr == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave][co * 8 + i] = s8[i];
    }
    __syncthreads();
    if (tid < 64) {
      float sm = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sm += red[w][tid];
      qinv[tid] = __builtin_amdgcn_rcpf(sm);
    }
  }
  // ---- B + C. chunks of LM_CH pixels
  const int ib = wave & 3, jb0 = 2 * (wave >> 2);
  f32x4 cacc[2] = {(f32x4)0.f, (f32x4)0.f};
  for (int n0 = 0; n0 < N; n0 += LM_CH) {
    __syncthreads();  // previous chunk's fragment reads are done (and qinv is written)
    // B: thread (wave, pr, co) handles pixels n0 + wave*8 + pr and + 64
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const int nl = hlf * 64 + wave * 8 + pr, n = n0 + nl;
      Vec8<f16> kv, vv, ko;
      if (n < N) {
        kv.load(kb + (long)n * qCs + co * 8);
        vv.load(vb + (long)n * qCs + co * 8);
        float e[8], m = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) m = fmaxf(m, kv.get(i));
        m = oct_max(m);
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { e[i] = __expf(kv.get(i) - m); sm += e[i]; }
        const float inv = __builtin_amdgcn_rcpf(oct_sum(sm));
#pragma unroll
        for (int i = 0; i < 8; ++i) ko.set(i, e[i] * inv);
      } else {
        ko.zero();
        vv.zero();
        (void)oct_max(0.f);  // keep the shuffles convergent for the lanes of a partially filled group
        (void)oct_sum(0.f);
      }
      ko.store(ks + nl * LM_LS + co * 8);
      vv.store(vs + nl * LM_LS + co * 8);
    }
    __syncthreads();
    // C: 4 k-steps of 32 pixels; A[i][n] = ks[n][i], B[n][j] = vs[n][j]: pixel index must be register-contiguous -> 2-byte gathers
#pragma unroll
    for (int kstep = 0; kstep < LM_CH / 32; ++kstep) {
      if (n0 + kstep * 32 >= N) break;  // uniform
      const f16* ap = ks + (kstep * 32 + 8 * g) * LM_LS + ib * 16 + r;
      Vec8<f16> a;
#pragma unroll
      for (int t = 0; t < 8; ++t) a.v[t] = ap[t * LM_LS];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const f16* bp = vs + (kstep * 32 + 8 * g) * LM_LS + (jb0 + jj) * 16 + r;
        Vec8<f16> bq;
#pragma unroll
        for (int t = 0; t < 8; ++t) bq.v[t] = bp[t * LM_LS];
        cacc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, bq.v, cacc[jj], 0, 0, 0);
      }
    }
  }
  // ---- D. ctx block (rows i = ib*16 + 4g + t, column j = jb*16 + r) -> ctxT[j][i]
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const f16x4 o = {(f16)cacc[jj][0], (f16)cacc[jj][1], (f16)cacc[jj][2], (f16)cacc[jj][3]};
    *reinterpret_cast<f16x4*>(ctxT + ((jb0 + jj) * 16 + r) * LM_LS + ib * 16 + 4 * g) = o;
  }
  __syncthreads();
  // ---- E. y[n][j] = sum_i qs[n][i] ctx[i][j]:  D[j][n] = A[j][i] B[i][n], A = ctxT (register resident), B = qs from q loads
  Vec8<f16> af[4][2];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb)
#pragma unroll
    for (int ksp = 0; ksp < 2; ++ksp) af[jb][ksp].load(ctxT + (jb * 16 + r) * LM_LS + ksp * 32 + 8 * g);
  float qm[2][8], qi[2][8];
#pragma unroll
  for (int ksp = 0; ksp < 2; ++ksp)
#pragma unroll
    for (int t = 0; t < 8; ++t) { qm[ksp][t] = qmax[ksp * 32 + 8 * g + t]; qi[ksp][t] = qinv[ksp * 32 + 8 * g + t]; }
  for (int blk = wave; blk * 16 < N; blk += 8) {
    const int n = blk * 16 + r;
    const bool ok = n < N;
    Vec8<f16> bq[2];
#pragma unroll
    for (int ksp = 0; ksp < 2; ++ksp) {
      Vec8<f16> q;
      if (ok) q.load(qb + (long)n * qCs + ksp * 32 + 8 * g);
      else q.zero();
#pragma unroll
      for (int t = 0; t < 8; ++t) bq[ksp].set(t, ok ? __expf(q.get(t) - qm[ksp][t]) * qi[ksp][t] : 0.f);
    }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      f32x4 acc = (f32x4)0.f;
#pragma unroll
      for (int ksp = 0; ksp < 2; ++ksp) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[jb][ksp].v, bq[ksp].v, acc, 0, 0, 0);
      if (ok) {
        const f16x4 o = {(f16)acc[0], (f16)acc[1], (f16)acc[2], (f16)acc[3]};
        *reinterpret_cast<f16x4*>(yb + (long)n * yCs + jb * 16 + 4 * g) = o;
      }
    }
  }
}

