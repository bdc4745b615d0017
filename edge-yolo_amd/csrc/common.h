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

static inline bool ey_aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }
static inline int ey_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

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

__device__ __forceinline__ float ey_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float ey_act(float x, int act) {
  switch (act) {
    case EY_ACT_SILU: return x * ey_sigmoid(x);
    case EY_ACT_RELU: return fmaxf(x, 0.f);
    case EY_ACT_SIGMOID: return ey_sigmoid(x);
    default: return x;
  }
}
