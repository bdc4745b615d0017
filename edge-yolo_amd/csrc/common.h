// Shared device/host helpers for libedgeyolo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/edgeyolo_hip.h"

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

int ey_set_error(int code, const char* fmt, ...);

#define EY_CHECK(cond, ...)                                  \
  do {                                                       \
    if (!(cond)) return ey_set_error(EY_EINVAL, __VA_ARGS__); \
  } while (0)

#define EY_LAUNCH_CHECK(name)                                                                  \
  do {                                                                                         \
    hipError_t e_ = hipGetLastError();                                                         \
    if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

// Row length (elements) of a packed conv weight row holding K = k*k*Cin values: +32 zero slack (masked tail lanes may read past the
// row), rounded so that the same pitch is bank-conflict-free for the MFMA fragment reads from LDS (lane (r, g) reads row r at +16g
// bytes with ds_read_b128, which the hardware serves in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, +32): f16 rows are
// 2 (mod 4) 16-byte units long -- an ODD number of units, the rule of rounds 1-2, is a 2-way conflict in every group
// (SQ_LDS_BANK_CONFLICT = 54 % of conv3_tile_kernel's LDS cycles; 0 at a pitch of 10 units, profiles/r03_conv3_kernels.txt) --
// f32 rows an odd number of 8-element (32-byte) units, i.e. also 2 (mod 4) 16-byte units.  A weight tile is one contiguous block in
// global memory (staging = flat copy, no per-vector row arithmetic).
static inline int ey_conv_kpad(int K, int es) {
  int units = (K + 32) >> 3;
  if (es == 2) while ((units & 3) != 2) ++units;
  else if (!(units & 1)) ++units;
  return units * 8;
}
static inline bool ey_aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }
static inline int ey_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence + barrier: the compiler puts `s_waitcnt vmcnt(0)` in front
// of it, so every barrier also waits for the wave's outstanding GLOBAL loads and stores -- the software-prefetched operands of the next
// tile, the epilogue stores of the last one (measured on conv3p_kernel: 57 % of the wave cycles parked in waits, MFMA pipe 16 % busy).
// Where the barrier only hands LDS contents between waves, this form keeps the vector-memory operations in flight across it (the
// hardware barrier itself does not drain them); registers loaded from global memory are still waited for at their first use.
__device__ __forceinline__ void ey_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2; MI355X_MICROARCH.md, speed only -- nothing here
// depends on it for correctness).  Kernels whose neighbouring blocks read overlapping input (stencil halos, window rows) take their work
// item from this permutation of the block index instead: every XCD then owns ONE contiguous range of the work list, and the shared
// lines are fetched into one L2 instead of two or three (measured, profiles/r03_pmc_traffic.json: 1.66-2.06x the algorithmic bytes
// with the plain index).
__device__ __forceinline__ unsigned ey_xcd_block(unsigned b, unsigned nb) {
  const unsigned q = nb >> 3, rem = nb & 7u, x = b & 7u, i = b >> 3;
  return x < rem ? x * (q + 1) + i : rem * (q + 1) + (x - rem) * q + i;
}

// ---- element <-> float
__device__ __forceinline__ float to_f(f16 v) { return (float)v; }
__device__ __forceinline__ float to_f(float v) { return v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ f16 from_f<f16>(float v) { return (f16)v; }
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }

// 8-element vector of T, loadable with 16-byte transactions
template <typename T> struct Vec8;
template <> struct Vec8<f16> {
  f16x8 v;
  __device__ __forceinline__ void load(const f16* p) { v = *reinterpret_cast<const f16x8*>(p); }
  __device__ __forceinline__ void store(f16* p) const { *reinterpret_cast<f16x8*>(p) = v; }
  __device__ __forceinline__ void zero() { v = (f16x8)(f16)0; }
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (f16)x; }
};
template <> struct Vec8<float> {
  f32x4 lo, hi;
  __device__ __forceinline__ void load(const float* p) {
    lo = *reinterpret_cast<const f32x4*>(p);
    hi = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ __forceinline__ void store(float* p) const {
    *reinterpret_cast<f32x4*>(p) = lo;
    *reinterpret_cast<f32x4*>(p + 4) = hi;
  }
  __device__ __forceinline__ void zero() { lo = (f32x4)0.f; hi = (f32x4)0.f; }
  __device__ __forceinline__ float get(int i) const { return i < 4 ? lo[i] : hi[i - 4]; }
  __device__ __forceinline__ void set(int i, float x) { if (i < 4) lo[i] = x; else hi[i - 4] = x; }
};

// acc[i] = fma(a[i], b[i], acc[i]) over the 8 f16 lanes of a and b with fp32 accumulation: v_fma_mix_f32 takes the f16 halves of its
// 32-bit sources directly (op_sel = which half, op_sel_hi = "this source is f16"), i.e. 8 instructions per vector pair instead of the
// 16 conversions + 8 FMAs the compiler emits for the same expression in the depthwise stencils (exactly the same arithmetic).
typedef unsigned int ey_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ey_fma8_mix(const Vec8<f16>& a, const Vec8<f16>& b, float (&acc)[8]) {
  const ey_u32x4 ua = __builtin_bit_cast(ey_u32x4, a.v), ub = __builtin_bit_cast(ey_u32x4, b.v);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * j]) : "v"(ua[j]), "v"(ub[j]));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * j + 1]) : "v"(ua[j]), "v"(ub[j]));
  }
}
__device__ __forceinline__ void ey_fma8_mix(const Vec8<float>& a, const Vec8<float>& b, float (&acc)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(a.get(i), b.get(i), acc[i]);
}

// 1/(1+e^-x) with the hardware reciprocal (1 ulp) instead of the IEEE division sequence (10+ instructions)
__device__ __forceinline__ float ey_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// x * sigmoid(x) as an fp32 value of its own.  Left to itself the compiler may fold the multiply into the conversion to f16 that
// follows (v_fma_mixlo_f16: ONE rounding) in one kernel and not in another (v_mul_f32 + v_cvt_f16_f32: two roundings).  The conv
// kernels round twice; the fused multi-layer kernels that promise the bits of their multi-launch form pin that here.
__device__ __forceinline__ float ey_silu_rn(float x) {
  float p = x * ey_sigmoid(x);
  asm("" : "+v"(p));
  return p;
}

// ---- buffer loads with hardware range checking: an out-of-range byte offset returns zeros, so padding taps,
// M tails and channel tails need no branches (offset EY_OOB is beyond any view: views are < 2 GiB, checked on the host)
#define EY_OOB 0x80000000u
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ey_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <typename T> struct BufLoad8;
template <> struct BufLoad8<f16> {
  // off: per-lane byte offset (VGPR), soff: wave-uniform byte offset (SGPR / immediate)
  static __device__ __forceinline__ void load(Vec8<f16>& v, __amdgpu_buffer_rsrc_t r, unsigned off, int soff = 0) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, soff, 0);
    v.v = __builtin_bit_cast(f16x8, t);
  }
};
template <> struct BufLoad8<float> {
  static __device__ __forceinline__ void load(Vec8<float>& v, __amdgpu_buffer_rsrc_t r, unsigned off, int soff = 0) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, soff, 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, soff + 16, 0);
    v.lo = __builtin_bit_cast(f32x4, a);
    v.hi = __builtin_bit_cast(f32x4, b);
  }
};
// The activation of N values with the (wave-uniform) selector tested ONCE: with ey_act() inside an unrolled loop the compiler keeps
// the scalar compare/branch chain per element, which also keeps the exp / rcp latencies of neighbouring elements from overlapping
// (measured: conv kernels 3-8 % faster with the SiLU loop on its own).
template <int N>
__device__ __forceinline__ void ey_act_n(float (&v)[N], int act);

__device__ __forceinline__ float ey_act(float x, int act) {
  switch (act) {
    case EY_ACT_SILU: return x * ey_sigmoid(x);
    case EY_ACT_RELU: return fmaxf(x, 0.f);
    case EY_ACT_SIGMOID: return ey_sigmoid(x);
    default: return x;
  }
}
template <int N>
__device__ __forceinline__ void ey_act_n(float (&v)[N], int act) {
  if (act == EY_ACT_SILU) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] * ey_sigmoid(v[i]);
  } else if (act == EY_ACT_RELU) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = fmaxf(v[i], 0.f);
  } else if (act == EY_ACT_SIGMOID) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ey_sigmoid(v[i]);
  }
}
