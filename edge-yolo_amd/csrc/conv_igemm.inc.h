// K1 — NHWC implicit-GEMM convolution on CDNA4 MFMA (gfx950), fused bias / activation / residual /
// bilinear-upsampled pre-activation add, virtual concat (+nearest x2) of up to two sources.
//
// GEMM view (per group):  D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],  k = (ky,kx,cin).
// We issue MFMA with A = weights, B = pixels ("swapped" orientation): the 16x16 accumulator then has the PIXEL on
// the lane (lane&15) and 4 consecutive MFMA rows in its 4 registers, so after the row permutation applied by
// ey_conv_pack_weight() every lane owns 4*NT CONSECUTIVE output channels of one pixel -> wide NHWC stores.
//
//   f16 : v_mfma_f32_16x16x32_f16   (lane: 8 consecutive k = 8 consecutive NHWC channels = one 16-byte load)
//   f32 : 8 x v_mfma_f32_16x16x4_f32 over the same 8-element fragments (exact f32 fma chain; parity mode)
//
// Workgroup = 4 waves; wave tile = (16*MT pixels) x (16*NT channels); block tile = 64*MT pixels x 16*NT channels.
// Pixels are fragment-shaped global loads (each wave owns its pixels; activations are read once because one block
// covers all of Cout up to 128); the weight tile shared by the 4 waves goes through LDS.
#include "common.h"
#include <type_traits>

struct ConvP {
  int B, H, W, Ho, Wo, Cout, k, stride, pad, act, nsrc;
  const void* src[2];
  int srcC[2], srcCs[2], srcUp[2];
  const void* w;
  const float* bias;
  void* y;
  int yCs;
  const void* res;
  int resCs;
  float out_scale;
  const void* addz;
  int addzCs, Hz, Wz;
  float zsy, zsx;  // input/output size ratios of the bilinear resize
  long srcG, yG;
  long wG;       // weight-set stride (elements) and highest set index: group g uses set min(g, wGmax)
  int wGmax;
  int Kpad;      // packed row length (elements)
  int nchunks;   // number of (tap, source, <=128-channel) K chunks
  unsigned srcBytes[2];  // byte extent of each source view (buffer-load range check)
  int Ctot, nsteps;      // total input channels; number of 32-channel K steps
  int NTpack;    // NT the weights were packed with (row permutation), see ey_conv_pack_weight
  int LSw;       // LDS row stride (elements) of the weight-stationary kernel
  long ntile;    // number of wave tiles (16*MT pixels each)
  int ntn;       // number of channel tiles (small-M kernel)
  int vec_store; // 1: y/res/addz views are aligned for 4-element vector access
  int tTR, tTC;  // stride-1 3x3 tile kernel: output tile rows x columns (tTR*tTC <= 256 pixels, (tTR+2)*(tTC+2) <= 340 halo pixels)
  int xcd;       // 3x3 stream / tile kernels: XCD-contiguous work order (ey_xcd_block)
};

__device__ __forceinline__ f32x4 mma16(const Vec8<f16>& a, const Vec8<f16>& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(const Vec8<float>& a, const Vec8<float>& b, f32x4 c) {
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[j], b.hi[j], c, 0, 0, 0);
  return c;
}

// K is walked in CHUNKS = (tap, source, <=128 channels).  Per chunk the block's weight tile [16*NT rows][chunk
// channels] is staged into LDS (register-staged: global loads are issued before the MFMAs of the current chunk and
// written to the other LDS buffer after them; padded rows -> conflict-free ds_read_b128), and every wave prefetches
// its own pixel fragments of the next chunk into registers.  One barrier per chunk.
#define CONV_CH 128           // channels per chunk
#define CONV_LS (CONV_CH + 8)  // LDS row stride in elements (odd multiple of 16 B for f16)

__device__ __forceinline__ void load4(const f16* p, float (&o)[4]) {
  const f16x4 v = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (float)v[j];
}
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
  const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = v[j];
}
__device__ __forceinline__ void store4(f16* p, const float* v) {
  const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  *reinterpret_cast<f16x4*>(p) = o;
}
__device__ __forceinline__ void store4(float* p, const float* v) {
  const f32x4 o = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p) = o;
}

template <typename T, int NT, int MT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* wlds = reinterpret_cast<T*>(smem);  // [2][16*NT][CONV_LS]
  constexpr int BN = 16 * NT;
  constexpr int WV = (BN * (CONV_CH / 8) + 255) / 256;  // 16-byte weight vectors staged per thread per chunk (max)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long M = (long)p.B * p.Ho * p.Wo;
  const long m_wave0 = ((long)blockIdx.x * 4 + wave) * (16 * MT);
  const int n_base = blockIdx.y * BN;
  const int grp = blockIdx.z;

  int pb[MT], poy[MT], pox[MT];
  bool pv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    long m = m_wave0 + mt * 16 + r;
    pv[mt] = m < M;
    long mm = pv[mt] ? m : 0;
    int hw = p.Ho * p.Wo;
    pb[mt] = (int)(mm / hw);
    int rem = (int)(mm - (long)pb[mt] * hw);
    poy[mt] = rem / p.Wo;
    pox[mt] = rem - poy[mt] * p.Wo;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4)0.f;

  const T* wg = (const T*)p.w + (long)min(grp, p.wGmax) * p.wG + (long)n_base * p.Kpad;  // this block's rows of the packed weights

  // chunk iterator (wave-uniform)
  int ky = 0, kx = 0, src = 0, c0 = 0, kofs = 0;
  auto chunk_cc = [&]() { return min(CONV_CH, p.srcC[src] - c0); };
  auto advance = [&]() {
    const int cc = chunk_cc();
    kofs += cc;
    c0 += cc;
    if (c0 >= p.srcC[src]) {
      c0 = 0;
      if (++src >= p.nsrc) {
        src = 0;
        if (++kx >= p.k) { kx = 0; ++ky; }
      }
    }
  };
  // issue the global loads of one chunk: weight vectors -> wst, this wave's pixel fragments -> px
  auto load_chunk = [&](Vec8<T> (&wst)[WV], Vec8<T> (&px)[MT][4]) {
    const int cc = chunk_cc(), ksteps = (cc + 31) >> 5, vpr = ksteps * 4;  // 16-byte vectors per staged row
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = threadIdx.x + i * 256;
      const int row = v / vpr, c8 = (v - row * vpr) * 8;
      if (row < BN && c8 < cc) wst[i].load(wg + (long)row * p.Kpad + kofs + c8);
      else wst[i].zero();
    }
    const int cs = p.srcCs[src], up = p.srcUp[src];
    const int Hs = p.H >> up, Ws = p.W >> up;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int iy = poy[mt] * p.stride - p.pad + ky, ix = pox[mt] * p.stride - p.pad + kx;
      const bool inb = pv[mt] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const long pix = ((long)pb[mt] * Hs + (inb ? (iy >> up) : 0)) * Ws + (inb ? (ix >> up) : 0);
      const T* xb = (const T*)p.src[src] + (long)grp * p.srcG + pix * cs + c0 + 8 * g;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < ksteps && inb && (ks * 32 + 8 * g) < cc) px[mt][ks].load(xb + ks * 32);
        else px[mt][ks].zero();
      }
    }
    return ksteps;
  };
  auto store_w = [&](const Vec8<T> (&wst)[WV], int ksteps, T* buf) {
    const int vpr = ksteps * 4;
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = threadIdx.x + i * 256;
      const int row = v / vpr, c8 = (v - row * vpr) * 8;
      if (row < BN) wst[i].store(buf + row * CONV_LS + c8);
    }
  };

  Vec8<T> wst[WV], cur[MT][4], nxt[MT][4];
  int ksteps = load_chunk(wst, cur);
  store_w(wst, ksteps, wlds);
  for (int ci = 0; ci < p.nchunks; ++ci) {
    __syncthreads();  // buffer ci&1 is complete; nobody still reads buffer (ci+1)&1
    const T* wb = wlds + (ci & 1) * (BN * CONV_LS);
    int ksteps_next = 0;
    const bool more = ci + 1 < p.nchunks;
    if (more) {
      advance();
      ksteps_next = load_chunk(wst, nxt);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < ksteps) {
        Vec8<T> af[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) af[nt].load(wb + (nt * 16 + r) * CONV_LS + ks * 32 + 8 * g);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16(af[nt], cur[mt][ks], acc[mt][nt]);
      }
    }
    if (more) {
      store_w(wst, ksteps_next, wlds + ((ci + 1) & 1) * (BN * CONV_LS));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cur[mt][ks] = nxt[mt][ks];
      ksteps = ksteps_next;
    }
  }

  // ---- epilogue: lane (pixel r of each m-block, group g) owns channels n_base + g*4NT + [0, 4NT)
  const int ch0 = n_base + g * 4 * NT;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (!pv[mt]) continue;
    const long m = m_wave0 + mt * 16 + r;
    float v[4 * NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * nt + j] = acc[mt][nt][j];
    if (p.bias) {
      const float* pbias = p.bias + min(grp, p.wGmax) * p.Cout;
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i)
        if (ch0 + i < p.Cout) v[i] += pbias[ch0 + i];
    }
    if (p.addz) {
      // F.interpolate(size=(Ho,Wo), bilinear, align_corners=False): ATen area_pixel_compute_source_index with
      // scale = in/out; exact x2 for even maps, the general ratio when the DWT floored an odd map
      const int Hz = p.Hz, Wz = p.Wz;
      float sy = fmaxf(p.zsy * (poy[mt] + 0.5f) - 0.5f, 0.f), sx = fmaxf(p.zsx * (pox[mt] + 0.5f) - 0.5f, 0.f);
      int y0 = (int)sy, x0 = (int)sx;
      int y1 = min(y0 + 1, Hz - 1), x1 = min(x0 + 1, Wz - 1);
      float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
      const T* z = (const T*)p.addz + (long)grp * p.yG;
      const T* z00 = z + (((long)pb[mt] * Hz + y0) * Wz + x0) * p.addzCs + ch0;
      const T* z01 = z + (((long)pb[mt] * Hz + y0) * Wz + x1) * p.addzCs + ch0;
      const T* z10 = z + (((long)pb[mt] * Hz + y1) * Wz + x0) * p.addzCs + ch0;
      const T* z11 = z + (((long)pb[mt] * Hz + y1) * Wz + x1) * p.addzCs + ch0;
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i)
        if (ch0 + i < p.Cout)
          v[i] += ly0 * (lx0 * to_f(z00[i]) + lx1 * to_f(z01[i])) + ly1 * (lx0 * to_f(z10[i]) + lx1 * to_f(z11[i]));
    }
    if (p.act == EY_ACT_SILU && p.out_scale == 1.f) {
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i) v[i] = v[i] * ey_sigmoid(v[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i) v[i] = ey_act(v[i], p.act) * p.out_scale;
    }
    T* yp = (T*)p.y + (long)grp * p.yG + m * p.yCs + ch0;
    const T* rp = p.res ? (const T*)p.res + (long)grp * p.yG + m * p.resCs + ch0 : nullptr;
    if (p.vec_store && ch0 + 4 * NT <= p.Cout) {
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        if (rp) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[4 * q + j] += to_f(rp[4 * q + j]);
        }
        if constexpr (sizeof(T) == 2) {
          f16x4 o = {(f16)v[4 * q], (f16)v[4 * q + 1], (f16)v[4 * q + 2], (f16)v[4 * q + 3]};
          *reinterpret_cast<f16x4*>(yp + 4 * q) = o;
        } else {
          f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
          *reinterpret_cast<f32x4*>(yp + 4 * q) = o;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i)
        if (ch0 + i < p.Cout) {
          float o = v[i] + (rp ? to_f(rp[i]) : 0.f);
          yp[i] = from_f<T>(o);
        }
    }
  }
}

// Shared epilogue: lane owns pixel m = (b, oy, ox) and output channels ch0 .. ch0+4NT of it.
template <typename T, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, const f32x4 (&acc)[NT], long m, int b, int oy, int ox, int ch0, int grp) {
  float v[4 * NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * nt + j] = acc[nt][j];
  const bool full = ch0 + 4 * NT <= p.Cout;
  if (p.bias) {
    const float* pbias = p.bias + min(grp, p.wGmax) * p.Cout;
    if (full && (p.wGmax == 0 || p.Cout % 4 == 0)) {
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(pbias + ch0 + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * q + j] += bv[j];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i)
        if (ch0 + i < p.Cout) v[i] += pbias[ch0 + i];
    }
  }
  if (p.addz) {
    // F.interpolate(size=(Ho,Wo), bilinear, align_corners=False): ATen area_pixel_compute_source_index, scale = in/out
    const int Hz = p.Hz, Wz = p.Wz;
    const float sy = fmaxf(p.zsy * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(p.zsx * (ox + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, Hz - 1), x1 = min(x0 + 1, Wz - 1);
    const float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const T* z = (const T*)p.addz + (long)grp * p.yG;
    const T* z00 = z + (((long)b * Hz + y0) * Wz + x0) * p.addzCs + ch0;
    const T* z01 = z + (((long)b * Hz + y0) * Wz + x1) * p.addzCs + ch0;
    const T* z10 = z + (((long)b * Hz + y1) * Wz + x0) * p.addzCs + ch0;
    const T* z11 = z + (((long)b * Hz + y1) * Wz + x1) * p.addzCs + ch0;
    if (p.vec_store && full) {
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        float a00[4], a01[4], a10[4], a11[4];
        load4(z00 + 4 * q, a00); load4(z01 + 4 * q, a01); load4(z10 + 4 * q, a10); load4(z11 + 4 * q, a11);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * q + j] += ly0 * (lx0 * a00[j] + lx1 * a01[j]) + ly1 * (lx0 * a10[j] + lx1 * a11[j]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i)
        if (ch0 + i < p.Cout)
          v[i] += ly0 * (lx0 * to_f(z00[i]) + lx1 * to_f(z01[i])) + ly1 * (lx0 * to_f(z10[i]) + lx1 * to_f(z11[i]));
    }
  }
  if (p.act == EY_ACT_SILU && p.out_scale == 1.f) {  // (wave-uniform; almost every conv of the network: one multiply per output less)
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) v[i] = v[i] * ey_sigmoid(v[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) v[i] = ey_act(v[i], p.act) * p.out_scale;
  }
  T* yp = (T*)p.y + (long)grp * p.yG + m * p.yCs + ch0;
  const T* rp = p.res ? (const T*)p.res + (long)grp * p.yG + m * p.resCs + ch0 : nullptr;
  if (p.vec_store && full) {
    if constexpr (sizeof(T) == 2 && NT % 2 == 0) {
      if (p.vec_store > 1) {  // 16-byte stores: a lane's 4*NT channels are contiguous (half the store instructions / sectors touched)
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {
          if (rp) {
            Vec8<T> rr;
            rr.load(rp + 8 * q);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[8 * q + j] += rr.get(j);
          }
          Vec8<T> o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o.set(j, v[8 * q + j]);
          o.store(yp + 8 * q);
        }
        return;
      }
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      if (rp) {
        float rr[4];
        load4(rp + 4 * q, rr);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * q + j] += rr[j];
      }
      store4(yp + 4 * q, v + 4 * q);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i)
      if (ch0 + i < p.Cout) yp[i] = from_f<T>(v[i] + (rp ? to_f(rp[i]) : 0.f));
  }
}

// ================================================================================================================
// Weight-stationary persistent variant (the default): each 512-thread workgroup stages its whole weight tile
// [16*NT rows][K] into LDS ONCE (zero-padded rows, odd 16-byte stride -> conflict-free ds_read_b128) and then its 8
// waves walk the pixel tiles independently (no barrier after the staging): pixel fragments come straight from
// global memory, up to 4 k-steps (128 channels) of loads in flight per wave, weights from LDS.  Weight traffic is
// per workgroup instead of per 128 pixels, and there is no per-chunk synchronisation.
template <typename T, int NT, int MT, int KS>
__global__ __launch_bounds__(512) void conv_ws_kernel(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* wl = reinterpret_cast<T*>(smem);
  constexpr int BN = 16 * NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n_base = blockIdx.y * BN;
  const int grp = blockIdx.z;
  {  // the weight tile is one contiguous block (LDS row stride == packed row stride): flat copy, 8 loads in flight per thread
    const T* wg = (const T*)p.w + (long)min(grp, p.wGmax) * p.wG + (long)n_base * p.Kpad;
    const int nvec = BN * (p.Kpad >> 3);
    for (int v0 = threadIdx.x; v0 < nvec; v0 += 512 * 8) {
      Vec8<T> w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v0 + u * 512 < nvec) w[u].load(wg + (long)(v0 + u * 512) * 8);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v0 + u * 512 < nvec) w[u].store(wl + (v0 + u * 512) * 8);
    }
  }
  __syncthreads();
  const long M = (long)p.B * p.Ho * p.Wo;
  const int hw = p.Ho * p.Wo;
  const T* wlane = wl + r * p.LSw + 8 * g;
  // channel run owned by this lane in the epilogue (row permutation of ey_conv_pack_weight with NTpack)
  const int BNp = 16 * p.NTpack;
  const int ch0 = (n_base / BNp) * BNp + g * 4 * p.NTpack + 4 * ((n_base % BNp) >> 4);

  for (long tile = (long)wave * gridDim.x + blockIdx.x; tile < p.ntile; tile += (long)gridDim.x * 8) {
    const long m0 = tile * (16 * MT);
    int pb[MT], poy[MT], pox[MT];
    bool pv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const long m = m0 + mt * 16 + r;
      pv[mt] = m < M;
      const long mm = pv[mt] ? m : 0;
      pb[mt] = (int)(mm / hw);
      const int rem = (int)(mm - (long)pb[mt] * hw);
      poy[mt] = rem / p.Wo;
      pox[mt] = rem - poy[mt] * p.Wo;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4)0.f;

    // per-source lane byte offsets of the window origin (tap 0,0) and the descriptors
    __amdgpu_buffer_rsrc_t rs[2];
    int base[2][MT];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s < p.nsrc) {
        rs[s] = ey_rsrc((const T*)p.src[s] + (long)grp * p.srcG, p.srcBytes[s]);
        const int up = p.srcUp[s], Hs = p.H >> up, Ws = p.W >> up;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int iy0 = poy[mt] * p.stride - p.pad, ix0 = pox[mt] * p.stride - p.pad;
          // up==1 only occurs with k==1 (pad 0): the origin itself is the (only) tap
          base[s][mt] = (((pb[mt] * Hs + (iy0 >> up)) * Ws + (ix0 >> up)) * p.srcCs[s] + 8 * g) * (int)sizeof(T);
        }
      }
    }
    int kofs = 0;
    if (KS == 3 && p.nsrc == 1 && p.srcC[0] <= 128) {
      // 3x3, one source of <= 128 channels (the stride-2 downsampling convs): a tap is ONE batch of <= 4 k-steps whose pixel fragments
      // come straight from global memory -- nine dependent round trips per tile when each batch is requested only after the previous
      // one was multiplied.  Here tap t+1 is in flight while tap t runs (two fragment buffers, statically alternated).
      const int Cs = p.srcC[0];
      Vec8<T> bq[2][MT][4];
      auto issue_tap = [&](int tap, Vec8<T> (&bf)[MT][4]) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int tapoff = (ky * p.W + kx) * p.srcCs[0] * (int)sizeof(T);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int iy = poy[mt] * p.stride - p.pad + ky, ix = pox[mt] * p.stride - p.pad + kx;
          const bool inb = pv[mt] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            BufLoad8<T>::load(bf[mt][ks], rs[0], (inb && (ks * 32 + 8 * g) < Cs) ? (unsigned)(base[0][mt] + tapoff + ks * 32 * (int)sizeof(T)) : EY_OOB);
        }
      };
      issue_tap(0, bq[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) issue_tap(tap + 1, bq[(tap + 1) & 1]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks * 32 < Cs) {
            Vec8<T> af[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) af[nt].load(wlane + nt * 16 * p.LSw + tap * Cs + ks * 32);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16(af[nt], bq[tap & 1][mt][ks], acc[mt][nt]);
          }
        }
      }
    } else {
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        bool inb[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int iy = poy[mt] * p.stride - p.pad + ky, ix = pox[mt] * p.stride - p.pad + kx;
          inb[mt] = pv[mt] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s < p.nsrc) {
            const int Cs = p.srcC[s];
            const int tapoff = (ky * p.W + kx) * p.srcCs[s] * (int)sizeof(T);  // (taps only exist when up == 0)
            unsigned voff[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) voff[mt] = inb[mt] ? (unsigned)(base[s][mt] + tapoff) : EY_OOB;
            for (int c0 = 0; c0 < Cs; c0 += 128) {
              const int rem = Cs - c0;  // channels left in this source
              Vec8<T> bf[MT][4];
              if (KS == 1 && rem >= 128) {
                // 1x1, full 128-channel batch (the common case): no per-k-step tests -- one basic block, so the LDS weight reads of step
                // k+1 are scheduled under the MFMAs of step k
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                  for (int mt = 0; mt < MT; ++mt) BufLoad8<T>::load(bf[mt][ks], rs[s], voff[mt] + (unsigned)((c0 + ks * 32) * (int)sizeof(T)));
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                  Vec8<T> af[NT];
#pragma unroll
                  for (int nt = 0; nt < NT; ++nt) af[nt].load(wlane + nt * 16 * p.LSw + kofs + c0 + ks * 32);
#pragma unroll
                  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16(af[nt], bf[mt][ks], acc[mt][nt]);
                }
              } else {
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) {
                if (ks * 32 < rem) {
                  const bool cok = (ks * 32 + 8 * g) < rem;
#pragma unroll
                  for (int mt = 0; mt < MT; ++mt)
                    BufLoad8<T>::load(bf[mt][ks], rs[s], cok ? voff[mt] + (unsigned)((c0 + ks * 32) * (int)sizeof(T)) : EY_OOB);
                }
              }
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) {
                if (ks * 32 < rem) {
                  Vec8<T> af[NT];
#pragma unroll
                  for (int nt = 0; nt < NT; ++nt) af[nt].load(wlane + nt * 16 * p.LSw + kofs + c0 + ks * 32);
#pragma unroll
                  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16(af[nt], bf[mt][ks], acc[mt][nt]);
                }
              }
              }
            }
            kofs += Cs;
          }
        }
      }
    }
    }

    // ---- epilogue
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (!pv[mt]) continue;
      conv_epilogue<T, NT>(p, acc[mt], m0 + mt * 16 + r, pb[mt], poy[mt], pox[mt], ch0, grp);
    }
  }
}

// ================================================================================================================
// 3x3 halo-tile variant (single source, Cin <= 64): the 9 taps of a 3x3 window re-read every input pixel up to nine
// times; from global memory that is 9x the L2->L1 traffic of the layer.  Here a 512-thread workgroup owns a tile of
// 8 output rows x 16*MT columns (wave w = row w): the (7S+3) x ((16MT-1)S+3) x Cin input halo goes to LDS once per
// tile (range-checked buffer loads: zero padding for free; the NEXT tile's halo is already in flight in registers
// while this one is multiplied), weights are LDS-resident for the whole kernel, and both MFMA operands are
// ds_read_b128.  Pixel rows are padded to an odd number of 16-byte units -> conflict-free fragment reads at stride 1.
template <typename T, int NT, int S>
__global__ __launch_bounds__(512) void conv3_halo_kernel(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BN = 16 * NT, MT = (S == 1) ? 2 : 1, TR = 8, TC = 16 * MT;
  constexpr int HR = (TR - 1) * S + 3, HC = (TC - 1) * S + 3;
  const int C = p.srcC[0], CV = C >> 3, CP = C + 8;
  constexpr int HV = 10;  // halo vectors per thread held in registers (>= HR*HC*CV/512 for Cin <= 64)
  T* wl = reinterpret_cast<T*>(smem);
  T* hl = wl + BN * p.LSw;  // [HR*HC][CP]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int n_base = blockIdx.y * BN, grp = blockIdx.z;
  {
    const T* wg = (const T*)p.w + (long)min(grp, p.wGmax) * p.wG + (long)n_base * p.Kpad;
    const int nvec = BN * (p.Kpad >> 3);  // contiguous block, LDS row stride == packed row stride
    for (int v0 = threadIdx.x; v0 < nvec; v0 += 512 * 8) {
      Vec8<T> w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v0 + u * 512 < nvec) w[u].load(wg + (long)(v0 + u * 512) * 8);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v0 + u * 512 < nvec) w[u].store(wl + (v0 + u * 512) * 8);
    }
  }
  const __amdgpu_buffer_rsrc_t rs = ey_rsrc((const T*)p.src[0] + (long)grp * p.srcG, p.srcBytes[0]);
  const int BNp = 16 * p.NTpack;
  const int ch0 = (n_base / BNp) * BNp + g * 4 * p.NTpack + 4 * ((n_base % BNp) >> 4);
  const int tilesX = (p.Wo + TC - 1) / TC, tilesY = (p.Ho + TR - 1) / TR, tiles_img = tilesX * tilesY;
  const long ntile = (long)p.B * tiles_img;
  const int nhv = HR * HC * CV;  // halo vectors per tile

  // this thread's halo vectors: LDS offset and (hy,hx,c8) are tile independent
  int hoff[HV], hyx[HV];
#pragma unroll
  for (int u = 0; u < HV; ++u) {
    const int v = threadIdx.x + u * 512;
    const int px = v / CV, cv = v - px * CV;
    hoff[u] = v < nhv ? px * CP + cv * 8 : -1;
    hyx[u] = ((px / HC) << 16) | ((px % HC) << 4) | 0;
    hyx[u] = (hyx[u] & ~0xF) | 0;  // (c8 recomputed from hoff)
  }
  auto issue_halo = [&](long tile, Vec8<T> (&hv)[HV]) {
    const int b = (int)(tile / tiles_img), tr = (int)(tile - (long)b * tiles_img);
    const int iy0 = (tr / tilesX) * TR * S - 1, ix0 = (tr % tilesX) * TC * S - 1;
#pragma unroll
    for (int u = 0; u < HV; ++u) {
      if (hoff[u] >= 0) {
        const int hy = hyx[u] >> 16, hx = (hyx[u] >> 4) & 0xFFF;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const int c8 = hoff[u] - (hy * HC + hx) * CP;
        BufLoad8<T>::load(hv[u], rs, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * p.srcCs[0] + c8) * (int)sizeof(T)) : EY_OOB);
      }
    }
  };

  Vec8<T> hv[HV];
  long tile = blockIdx.x;
  if (tile < ntile) issue_halo(tile, hv);
  const T* wlane = wl + r * p.LSw + 8 * g;
  for (; tile < ntile; tile += gridDim.x) {
    __syncthreads();  // weights staged (first pass) / every wave finished reading the previous halo
#pragma unroll
    for (int u = 0; u < HV; ++u)
      if (hoff[u] >= 0) hv[u].store(hl + hoff[u]);
    __syncthreads();
    const long nxt = tile + gridDim.x;
    if (nxt < ntile) issue_halo(nxt, hv);  // in flight during the MFMAs below

    const int b = (int)(tile / tiles_img), tr = (int)(tile - (long)b * tiles_img);
    const int oy = (tr / tilesX) * TR + wave, ox0 = (tr % tilesX) * TC;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4)0.f;
    const T* hlane = hl + ((wave * S) * HC + r * S) * CP + 8 * g;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        for (int c0 = 0; c0 < C; c0 += 32) {
          const bool cok = (c0 + 8 * g) < C;
          Vec8<T> bf[MT], af[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (cok) bf[mt].load(hlane + (ky * HC + mt * 16 * S + kx) * CP + c0);
            else bf[mt].zero();
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) af[nt].load(wlane + nt * 16 * p.LSw + (ky * 3 + kx) * C + c0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16(af[nt], bf[mt], acc[mt][nt]);
        }
      }
    }
    if (oy < p.Ho) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int ox = ox0 + mt * 16 + r;
        if (ox < p.Wo) conv_epilogue<T, NT>(p, acc[mt], ((long)b * p.Ho + oy) * p.Wo + ox, b, oy, ox, ch0, grp);
      }
    }
  }
}

// ================================================================================================================
// Small-M variant (feature maps of 40x40 and below: M = B*Ho*Wo is a few 10^4 pixels).  Such layers are pure latency:
// a kernel that first stages weights into LDS, synchronises, then loads pixels pays 4-5 dependent memory round trips
// for a few microseconds of work.  Here a wave owns one 16-pixel block x 16*NT channels, there is no LDS and no
// barrier, and the fragment loads of up to BATCH k-steps (pixels by range-checked buffer loads, weights straight
// from the L2-resident packed array) are all in flight before the first MFMA: one round trip per BATCH*32 channels.
template <typename T, int NT, int BATCH>
__global__ __launch_bounds__(256) void conv_small_kernel(ConvP p) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int ntn = p.ntn;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_tile = (int)(wid % ntn);
  const long m_tile = wid / ntn;
  if (m_tile >= p.ntile) return;  // whole wave
  const int grp = blockIdx.z;
  const int n_base = n_tile * 16 * NT;
  const long M = (long)p.B * p.Ho * p.Wo;
  const long m = m_tile * 16 + r;
  const bool pv = m < M;
  const int hw = p.Ho * p.Wo;
  const long mm = pv ? m : 0;
  const int b = (int)(mm / hw), rem = (int)(mm - (long)b * hw), oy = rem / p.Wo, ox = rem - oy * p.Wo;
  const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;

  const __amdgpu_buffer_rsrc_t rs0 = ey_rsrc((const T*)p.src[0] + (long)grp * p.srcG, p.srcBytes[0]);
  const __amdgpu_buffer_rsrc_t rs1 = p.nsrc > 1 ? ey_rsrc(p.src[1], p.srcBytes[1]) : rs0;
  int base[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int up = p.srcUp[s], Hs = p.H >> up, Ws = p.W >> up;  // up==1 only with k==1
    base[s] = (((b * Hs + (iy0 >> up)) * Ws + (ix0 >> up)) * p.srcCs[s] + 8 * g) * (int)sizeof(T);
  }
  unsigned tapmask = 0;
  for (int t = 0; t < p.k * p.k; ++t) {
    const int iy = iy0 + t / p.k, ix = ix0 + t % p.k;
    tapmask |= (unsigned)(pv && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) << t;
  }
  const T* wl = (const T*)p.w + (long)min(grp, p.wGmax) * p.wG + (long)(n_base + r) * p.Kpad + 8 * g;

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4)0.f;

  int s_ = 0, tap = 0, c = 0;  // wave-uniform K cursor: (tap, source, channel)
  for (int step0 = 0; step0 < p.nsteps; step0 += BATCH) {
    Vec8<T> bf[BATCH], af[BATCH][NT];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      if (step0 + u < p.nsteps) {
        const int ky = tap / p.k, kx = tap - ky * p.k;
        const int Cs = p.srcC[s_];
        const bool ok = ((tapmask >> tap) & 1u) && (c + 8 * g) < Cs;
        const unsigned off = ok ? (unsigned)(base[s_] + ((ky * p.W + kx) * p.srcCs[s_] + c) * (int)sizeof(T)) : EY_OOB;
        if (s_) BufLoad8<T>::load(bf[u], rs1, off);
        else BufLoad8<T>::load(bf[u], rs0, off);
        const int kofs = tap * p.Ctot + (s_ ? p.srcC[0] : 0) + c;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) af[u][nt].load(wl + (long)nt * 16 * p.Kpad + kofs);
        c += 32;
        if (c >= Cs) {
          c = 0;
          if (++s_ >= p.nsrc) { s_ = 0; ++tap; }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      if (step0 + u < p.nsteps) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mma16(af[u][nt], bf[u], acc[nt]);
      }
    }
  }
  if (!pv) return;
  const int BNp = 16 * p.NTpack;
  const int ch0 = (n_base / BNp) * BNp + g * 4 * p.NTpack + 4 * ((n_base % BNp) >> 4);
  conv_epilogue<T, NT>(p, acc, m, b, oy, ox, ch0, grp);
}

// ================================================================================================================
// 3x3 tile kernel (single source, any Cin % 8 == 0, stride 1 or 2): the GEMM-shaped 3x3 convs (K = 9*Cin up to 2304) want a
// big register tile per wave, which the weight-stationary kernel cannot give them (its [16*NT][K] weight tile must fit LDS,
// so NT shrinks to 1-2 and every MFMA needs its own operand loads).  Here a 256-thread workgroup owns 8 output rows x 32
// (stride 1) / 16 (stride 2) columns and 16*NT <= 64 output channels; a wave = 2 rows = MT (4 / 2) pixel blocks x NT channel
// blocks, i.e. MT*NT MFMAs per MT + NT operand fragments.  K is walked in 32-channel chunks: the input halo of the chunk
// goes to LDS once (range-checked buffer loads = zero padding; the next chunk's halo is already in flight in registers)
// and is shared by all 9 taps and all waves; the weight fragments come straight from the L2-resident packed array by
// scalar-offset buffer loads (one instruction each, double-buffered per tap) -- no weight staging, no weight LDS, so
// several workgroups fit a CU and hide each other's latencies.  Stride 2 stores the halo split by column parity so that
// fragment reads stay at the conflict-free odd 16-byte pixel stride.
// WLDS: the chunk's [9 taps][16*NT rows][32 channels] weight block is staged into LDS next to the halo (one global read per
// workgroup instead of one per wave: measured, the per-wave weight fragment loads saturate the CU's 64 B/clk vector-memory
// path long before the MFMA pipe; LDS delivers 256 B/clk) -- at the price of 46 KB more LDS per workgroup.
template <typename T, int NT, int S, bool WLDS>
__global__ __launch_bounds__(256) void conv3_tile_kernel(ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TR = 8, MTC = (S == 1) ? 2 : 1, TC = 16 * MTC, MT = 2 * MTC;
  constexpr int HR = (TR - 1) * S + 3, HC = (TC - 1) * S + 3, HCH = (HC + 1) / 2, CC = 32, CP = CC + 8;
  constexpr int LROW = (S == 1) ? HC : 2 * HCH;            // LDS pixels per halo row
  constexpr int NHV = (HR * HC * (CC / 8) + 255) / 256;    // halo vectors per thread
  constexpr int NWVEC = 9 * 16 * NT * (CC / 8), NWV = WLDS ? (NWVEC + 255) / 256 : 1;  // weight vectors per chunk / per thread
  T* hl = reinterpret_cast<T*>(smem);                      // [HR][LROW][CP]
  T* wl = hl + HR * LROW * CP;                             // [9][16*NT][CP]   (WLDS)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int grp = blockIdx.z, n_base = blockIdx.y * (16 * NT);
  // stride 1: the tile is tTR x tTC output pixels FLATTENED onto the 256 pixel slots (slot q = row*tTC + col), chosen on the host so
  // that small maps are covered with little waste (20x20 -> 12x20 tiles: 78 % of the slots do useful work instead of 52 % with 8x32);
  // stride 2: the fixed 8 x 16 tile with the parity-split halo.
  const int tTR = (S == 1) ? p.tTR : TR, tTC = (S == 1) ? p.tTC : TC;
  const int hHR = (S == 1) ? tTR + 2 : HR, hHC = (S == 1) ? tTC + 2 : HC, lrow = (S == 1) ? hHC : LROW;
  const float inv_hc = 1.0f / (float)hHC, inv_tc = 1.0f / (float)tTC;
  const int tilesX = (p.Wo + tTC - 1) / tTC, tilesY = (p.Ho + tTR - 1) / tTR, tiles_img = tilesX * tilesY;
  const int bid = p.xcd ? (int)ey_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;  // neighbouring tiles (shared halos) in one XCD's L2
  const int b = bid / tiles_img, trem = bid - b * tiles_img;
  const int oy0 = (trem / tilesX) * tTR, ox0 = (trem % tilesX) * tTC;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  const int Cin = p.srcC[0];
  const __amdgpu_buffer_rsrc_t rs = ey_rsrc((const T*)p.src[0] + (long)grp * p.srcG, p.srcBytes[0]);
  const __amdgpu_buffer_rsrc_t rw = ey_rsrc((const T*)p.w + (long)min(grp, p.wGmax) * p.wG + (long)n_base * p.Kpad, (unsigned)(16 * NT * p.Kpad * (int)sizeof(T)));
  const unsigned wvoff = (unsigned)((r * p.Kpad + 8 * g) * (int)sizeof(T));
  const int rowblk = 16 * p.Kpad * (int)sizeof(T);

  // this thread's halo vectors: global byte offset (chunk 0) or OOB, and LDS element offset (tile independent)
  unsigned hgo[NHV];
  int hlo[NHV];
  int hcv[NHV];
#pragma unroll
  for (int u = 0; u < NHV; ++u) {
    const int v = threadIdx.x + u * 256;
    // (runtime divisor: exact float-reciprocal quotient for these small ranges, 3 instructions instead of a ~35-instruction division)
    const int px = v >> 2, cv = v & 3, hy = (S == 1) ? (int)(((float)px + 0.5f) * inv_hc) : px / HC, hx = px - hy * hHC;
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool ok = v < hHR * hHC * 4 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    hgo[u] = ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * p.srcCs[0] + cv * 8) * (int)sizeof(T)) : EY_OOB;
    const int lpix = (S == 1) ? hy * lrow + hx : hy * LROW + (hx & 1) * HCH + (hx >> 1);
    hlo[u] = v < hHR * hHC * 4 ? lpix * CP + cv * 8 : -1;
    hcv[u] = cv * 8;
  }
  auto issue_halo = [&](int c0, Vec8<T> (&hv)[NHV]) {
#pragma unroll
    for (int u = 0; u < NHV; ++u) BufLoad8<T>::load(hv[u], rs, (c0 + hcv[u]) < Cin ? hgo[u] : EY_OOB, c0 * (int)sizeof(T));
  };
  // fragment base of this lane in the halo: wave owns output rows 2*wave, 2*wave+1; block mt = (row mt/MTC, column block mt%MTC)
  int bbase[MT], prow[MT], pcol[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if constexpr (S == 1) {
      const int q = wave * 64 + mt * 16 + r;  // pixel slot
      const bool qv = q < tTR * tTC;
      prow[mt] = qv ? (int)(((float)q + 0.5f) * inv_tc) : 0;
      pcol[mt] = qv ? q - prow[mt] * tTC : 0;
      bbase[mt] = (prow[mt] * lrow + pcol[mt]) * CP + 8 * g;
      if (!qv) prow[mt] = 1 << 20;  // never stored
    } else {
      prow[mt] = 2 * wave + mt / MTC;
      pcol[mt] = (mt % MTC) * 16 + r;
      bbase[mt] = (prow[mt] * 2 * LROW + pcol[mt]) * CP + 8 * g;
    }
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4)0.f;

  Vec8<T> hv[NHV];
  issue_halo(0, hv);
  if constexpr (WLDS) {
    // this thread's weight vectors of a chunk: vector v = ((tap*16NT + row)*4 + cv)
    unsigned wgo[NWV];
#pragma unroll
    for (int u = 0; u < NWV; ++u) {
      const int v = threadIdx.x + u * 256, cv = v & 3, rt = v >> 2, tap = rt / (16 * NT), row = rt - tap * (16 * NT);
      wgo[u] = v < NWVEC ? (unsigned)((row * p.Kpad + tap * Cin + cv * 8) * (int)sizeof(T)) : EY_OOB;
    }
    Vec8<T> wv[NWV];
    auto issue_w = [&](int c0) {
#pragma unroll
      for (int u = 0; u < NWV; ++u) BufLoad8<T>::load(wv[u], rw, wgo[u], c0 * (int)sizeof(T));
    };
    issue_w(0);
    const T* wlane = wl + r * CP + 8 * g;
    for (int c0 = 0; c0 < Cin; c0 += CC) {
      __syncthreads();  // every wave finished reading the previous chunk's halo and weights
#pragma unroll
      for (int u = 0; u < NHV; ++u)
        if (hlo[u] >= 0) hv[u].store(hl + hlo[u]);
#pragma unroll
      for (int u = 0; u < NWV; ++u) {
        const int v = threadIdx.x + u * 256;
        if (v < NWVEC) wv[u].store(wl + (v >> 2) * CP + (v & 3) * 8);
      }
      __syncthreads();
      if (c0 + CC < Cin) {  // next chunk: in flight during the 9 taps below
        issue_halo(c0 + CC, hv);
        issue_w(c0 + CC);
      }
      // operand fragments double-buffered across taps: tap t+1's 8 LDS reads are issued before tap t's 16 MFMAs
      Vec8<T> bfr[2][MT], afr[2][NT];
      auto ld_tap = [&](int tap, Vec8<T> (&a)[NT], Vec8<T> (&bq)[MT]) {
        const int ky = tap / 3, kx = tap % 3;
        const int toff = ((S == 1) ? (ky * lrow + kx) : (ky * LROW + (kx & 1) * HCH + (kx >> 1))) * CP;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[nt].load(wlane + (tap * 16 * NT + nt * 16) * CP);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) bq[mt].load(hl + bbase[mt] + toff);
      };
      ld_tap(0, afr[0], bfr[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap < 8) ld_tap(tap + 1, afr[(tap + 1) & 1], bfr[(tap + 1) & 1]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16(afr[tap & 1][nt], bfr[tap & 1][mt], acc[mt][nt]);
      }
    }
  } else {
  // weight fragments straight from L2: ring of 4 tap buffers, 3 taps ahead of the MFMAs
  Vec8<T> af[4][NT];
  auto load_w = [&](int tap, int c0, Vec8<T> (&a)[NT]) {
    const int kofs = (tap * Cin + c0) * (int)sizeof(T);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) BufLoad8<T>::load(a[nt], rw, wvoff, nt * rowblk + kofs);
  };
  load_w(0, 0, af[0]);
  load_w(1, 0, af[1]);
  load_w(2, 0, af[2]);
  for (int c0 = 0; c0 < Cin; c0 += CC) {
    __syncthreads();  // every wave finished reading the previous chunk's halo
#pragma unroll
    for (int u = 0; u < NHV; ++u)
      if (hlo[u] >= 0) hv[u].store(hl + hlo[u]);
    __syncthreads();
    const bool more = c0 + CC < Cin;
    if (more) issue_halo(c0 + CC, hv);  // in flight during the 9 taps below
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      if (tap + 3 < 9) load_w(tap + 3, c0, af[(tap + 3) & 3]);
      else if (more) load_w(tap + 3 - 9, c0 + CC, af[(tap + 3) & 3]);
      const int toff = ((S == 1) ? (ky * lrow + kx) : (ky * LROW + (kx & 1) * HCH + (kx >> 1))) * CP;
      Vec8<T> bf[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bf[mt].load(hl + bbase[mt] + toff);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16(af[tap & 3][nt], bf[mt], acc[mt][nt]);
    }
    // 9 taps per chunk: the next chunk's taps 0,1,2 sit in buffers 1,2,3 -> rotate so that every chunk starts on buffer 0
    if (more) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { af[0][nt] = af[1][nt]; af[1][nt] = af[2][nt]; af[2][nt] = af[3][nt]; }
    }
  }
  }
  const int BNp = 16 * p.NTpack;
  const int ch0 = (n_base / BNp) * BNp + g * 4 * p.NTpack + 4 * ((n_base % BNp) >> 4);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int oy = oy0 + prow[mt], ox = ox0 + pcol[mt];
    if (oy < p.Ho && ox < p.Wo) conv_epilogue<T, NT>(p, acc[mt], ((long)b * p.Ho + oy) * p.Wo + ox, b, oy, ox, ch0, grp);
  }
}

// ================================================================================================================
// 3x3 persistent tile kernel for Cin = 64, stride 1 (f16): the Detect box-tower convs (64->64 at 80x80 / 40x40 / 20x20) and the
// other K = 576 stride-1 convs.  Phase ablation of this kernel's first version (profiles/r03_conv3p_ablation.txt): of 37 us at 80x80,
// batch 32, the EPILOGUE was ~20 us and the K loop ~12 -- 13 M outputs x (bias, v_exp, v_rcp, 2 multiplies, convert) on one wave per
// SIMD with nothing beside it, then 16-byte stores at a 32-byte stride; the tile kernel above pays the same, plus a restaged 9-tap
// weight block per 32-channel chunk of every tile between two barriers per chunk.  Here:
//   * one 256-thread workgroup per CU, persistent over its tiles; the whole [9][64][64] weight block of its channel tile sits in LDS
//     for the life of the workgroup (staged once) next to ONE all-channel halo buffer (<= 340 pixels x 64 ch): a tile is 18 k-steps
//     x 16 MFMAs per wave between two LDS-only barriers (ey_lds_barrier: the vector-memory queue is not drained), the operand
//     fragments of step k + 1 are read while step k multiplies, the next tile's halo is requested before the K loop;
//   * FAST (bias + SiLU, no residual / resize-add, 16-byte-aligned full channel tile: every conv this kernel is dispatched for in the
//     network): the EPILOGUE IS DEFERRED AND INTERLEAVED -- the finished accumulators of tile i move to a second register set and
//     are activated, converted and stored in 16 branch-free quarters (4 outputs per lane each) placed behind k-steps 0..15 of tile
//     i + 1, and sched_group_barrier pins "1 MFMA, 2 VALU" so the v_exp / v_rcp chains issue in the MFMAs' shadow (an MFMA holds
//     the vector issue port 8 of its 16 cycles).  Stores are range-checked buffer stores (partial tiles / first tile: offset out of
//     range -> dropped by the hardware), so the K loop stays ONE basic block.  Same arithmetic as conv_epilogue (bit-identical).
//   * row pitch of both LDS arrays = 80 halves = 10 16-byte units: 2 (mod 4) units is conflict-free for ds_read_b128's lane groups
//     (SQ_LDS_BANK_CONFLICT = 0; the ODD pitch of the older kernels is a 2-way conflict: 54 % of the tile kernel's LDS cycles).
// Tile geometry = the flattened tTR x tTC pixel slots of conv3_tile_kernel (chosen on the host per map size).
template <int NT, bool FAST>
__global__ __launch_bounds__(256, 1) void conv3p_kernel(ConvP p) {
  typedef f16 T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MT = 4, CP = 80, NHV = (340 * 8 + 255) / 256;
  T* wl = reinterpret_cast<T*>(smem);            // [9][16*NT][CP]
  T* hl = wl + 9 * 16 * NT * CP;                 // [<= 340 halo pixels][CP]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int n_base = blockIdx.y * (16 * NT);
  const int tTR = p.tTR, tTC = p.tTC, hHR = tTR + 2, hHC = tTC + 2;
  const float inv_hc = 1.0f / (float)hHC, inv_tc = 1.0f / (float)tTC;
  const int tilesX = (p.Wo + tTC - 1) / tTC, tilesY = (p.Ho + tTR - 1) / tTR, tiles_img = tilesX * tilesY;
  const int ntiles = p.B * tiles_img;  // (< 2^31: checked on the host)
  const __amdgpu_buffer_rsrc_t rs = ey_rsrc(p.src[0], p.srcBytes[0]);
  {  // weights: packed rows [n_base + row][tap * 64 + c] -> LDS [tap][row][CP]; every vector requested before the first LDS store
    const T* wg = (const T*)p.w + (long)n_base * p.Kpad;
    constexpr int NWV = 9 * 16 * NT * 8 / 256;
    Vec8<T> w[NWV];
#pragma unroll
    for (int u = 0; u < NWV; ++u) {
      const int v = threadIdx.x + u * 256, cv = v & 7, row = (v >> 3) % (16 * NT), tap = (v >> 3) / (16 * NT);
      w[u].load(wg + (long)row * p.Kpad + tap * 64 + cv * 8);
    }
#pragma unroll
    for (int u = 0; u < NWV; ++u) {
      const int v = threadIdx.x + u * 256, cv = v & 7, row = (v >> 3) % (16 * NT), tap = (v >> 3) / (16 * NT);
      w[u].store(wl + (tap * 16 * NT + row) * CP + cv * 8);
    }
  }
  // halo vectors of this thread: LDS element offset (tile independent) and (row, column) inside the halo
  int hlo[NHV], hyx[NHV];
#pragma unroll
  for (int u = 0; u < NHV; ++u) {
    const int v = threadIdx.x + u * 256, px = v >> 3, cv = v & 7;
    const int hy = (int)(((float)px + 0.5f) * inv_hc), hx = px - hy * hHC;
    hlo[u] = v < hHR * hHC * 8 ? px * CP + cv * 8 : -1;
    hyx[u] = (hy << 16) | hx;
  }
  Vec8<T> hv[NHV];
  auto issue_halo = [&](int tile) {
    const int b = tile / tiles_img, trem = tile - b * tiles_img;
    const int iy0 = (trem / tilesX) * tTR - 1, ix0 = (trem % tilesX) * tTC - 1;
#pragma unroll
    for (int u = 0; u < NHV; ++u) {
      const int iy = iy0 + (hyx[u] >> 16), ix = ix0 + (hyx[u] & 0xFFFF);
      const bool ok = (hlo[u] >= 0) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
      BufLoad8<T>::load(hv[u], rs, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * p.srcCs[0] + (threadIdx.x & 7) * 8) * 2) : EY_OOB, 0);
    }
  };
  // fragment bases of this lane: pixel slot q = wave * 64 + mt * 16 + r of the flattened tile
  int bbase[MT], prow[MT], pcol[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int q = wave * 64 + mt * 16 + r;
    const bool qv = q < tTR * tTC;
    prow[mt] = qv ? (int)(((float)q + 0.5f) * inv_tc) : 0;
    pcol[mt] = qv ? q - prow[mt] * tTC : 0;
    bbase[mt] = (prow[mt] * hHC + pcol[mt]) * CP + 8 * g;
    if (!qv) prow[mt] = 1 << 20;  // never stored
  }
  const T* wlane = wl + r * CP + 8 * g;
  const int BNp = 16 * p.NTpack;
  const int ch0 = (n_base / BNp) * BNp + g * 4 * p.NTpack + 4 * ((n_base % BNp) >> 4);

  // Two accumulator sets alternate between tiles (FAST): the MFMAs of tile i write set i & 1 while the epilogue quarters read set
  // (i - 1) & 1 -- both stay in the accumulator registers, a quarter pulls its 4 values out when it runs (no second VGPR copy of a
  // whole tile: the arch-VGPR budget goes to the double-buffered operand fragments instead).
  f32x4 acc[2][MT][NT];
  // FAST epilogue state: bias of this lane's 4*NT channels, the output view as a range-checked buffer, the previous tile's per-block
  // store offsets (out of range = dropped: pixels outside the map, and everything while there is no previous tile)
  float bz[4 * NT];
  unsigned yoff[MT];
  unsigned oh[2 * NT];  // a block's 4*NT converted outputs, two halves per register
  __amdgpu_buffer_rsrc_t ry = ey_rsrc(p.y, p.srcBytes[1]);
  if constexpr (FAST) {
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + ch0 + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) bz[4 * q + j] = bv[j];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) yoff[mt] = EY_OOB;
  }
  auto set_yoff = [&](int t) {
    const int b = t / tiles_img, trem = t - b * tiles_img;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int oy = (trem / tilesX) * tTR + prow[mt], ox = (trem % tilesX) * tTC + pcol[mt];
      yoff[mt] = (oy < p.Ho && ox < p.Wo) ? (unsigned)(((((long)b * p.Ho + oy) * p.Wo + ox) * p.yCs + ch0) * 2) : EY_OOB;
    }
  };
  auto quarter = [&](const f32x4 (&src)[MT][NT], int mt, int q) {  // outputs 4q .. 4q+3 of block mt of a finished tile (conv_epilogue's arithmetic)
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = src[mt][q][j] + bz[4 * q + j];
      v[j] = v[j] * ey_sigmoid(v[j]);
    }
    const f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    const uint2 u = __builtin_bit_cast(uint2, h);
    oh[2 * q] = u.x;
    oh[2 * q + 1] = u.y;
    if (q == NT - 1) {
#pragma unroll
      for (int c = 0; c < NT / 2; ++c) {
        const u32x4 d = {oh[4 * c], oh[4 * c + 1], oh[4 * c + 2], oh[4 * c + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(d, ry, (int)yoff[mt], 16 * c, 0);
      }
    }
  };
  auto tile_body = [&](auto cur_c, int tile) {
    constexpr int CUR = decltype(cur_c)::value;
    ey_lds_barrier();  // every wave finished reading the previous tile's halo (first pass: the weights are staged)
#pragma unroll
    for (int u = 0; u < NHV; ++u)
      if (hlo[u] >= 0) hv[u].store(hl + hlo[u]);
    ey_lds_barrier();  // (LDS only: neither the halo prefetch below nor the epilogue stores are drained at a barrier)
    if (tile + gridDim.x < ntiles) issue_halo(tile + gridDim.x);  // in flight during the K loop below
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[CUR][mt][nt] = (f32x4)0.f;
    // 18 k-steps; the operand fragments of step k + 1 are read from LDS while the 16 MFMAs of step k are issued
    Vec8<T> af[2][NT], bf[2][MT];
    auto ld_step = [&](int k, Vec8<T> (&a)[NT], Vec8<T> (&bq)[MT]) {
      const int tap = k >> 1, ks = k & 1;
      const int toff = ((tap / 3) * hHC + tap % 3) * CP + ks * 32;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) a[nt].load(wlane + (tap * 16 * NT + nt * 16) * CP + ks * 32);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bq[mt].load(hl + bbase[mt] + toff);
    };
    ld_step(0, af[0], bf[0]);
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      // step k + 1's fragments are requested HERE, a whole step (16 MFMAs = 256 cycles) before their first use: the scheduling barrier
      // keeps the scheduler from sinking the reads to the MFMAs that consume them (it does, left alone, and every step then waits
      // out an LDS round trip: 33 us -> see profiles/r03_c3s_c3p_vs_tile.txt)
      if (k < 17) ld_step(k + 1, af[(k + 1) & 1], bf[(k + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[CUR][mt][nt] = mma16(af[k & 1][nt], bf[k & 1][mt], acc[CUR][mt][nt]);
      if constexpr (FAST) {
        if (k < MT * NT) quarter(acc[1 - CUR], k / NT, k % NT);  // one quarter-block of the previous tile's epilogue per k-step
        // pinned issue order: every MFMA of the step with two epilogue VALU ops in its shadow
#pragma unroll
        for (int i = 0; i < MT * NT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FAST) {
      set_yoff(tile);
    } else {
      const int b = tile / tiles_img, trem = tile - b * tiles_img;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int oy = (trem / tilesX) * tTR + prow[mt], ox = (trem % tilesX) * tTC + pcol[mt];
        if (oy < p.Ho && ox < p.Wo) conv_epilogue<T, NT>(p, acc[CUR][mt], ((long)b * p.Ho + oy) * p.Wo + ox, b, oy, ox, ch0, 0);
      }
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) issue_halo(tile);
  int last = -1;  // accumulator set of the last finished tile (workgroup-uniform)
  while (tile < ntiles) {
    tile_body(std::integral_constant<int, 0>{}, tile);
    last = 0;
    tile += gridDim.x;
    if (tile >= ntiles) break;
    tile_body(std::integral_constant<int, 1>{}, tile);
    last = 1;
    tile += gridDim.x;
  }
  if constexpr (FAST) {  // the last tile's epilogue
    if (last == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < NT; ++q) quarter(acc[0], mt, q);
    } else if (last == 1) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < NT; ++q) quarter(acc[1], mt, q);
    }
  }
}

// ================================================================================================================
// 3x3 "stream" kernel (f16, one source, Cin = 64 * UPT, stride 1 or 2): weight-stationary like conv_ws_kernel, but shaped for the
// K = 576 ... 2304 convs (backbone down-sampling layers 3/5/7/17/20, the Detect box towers) whose time the tile kernel above spends
// outside the MFMA loop (halo staging and two barriers per 32-channel chunk, a prologue and an epilogue every workgroup reaches at
// the same moment).  Here a 512-thread workgroup stages its [16*NT][K] weight tile into LDS ONCE and its 8 waves then run
// independently for the life of the grid -- no barrier after the staging, so the waves drift apart and one wave's epilogue (SiLU on
// the VALU, stores) runs beside its SIMD partner's MFMAs:
//   * wave tile = 16*MT consecutive output pixels (flattened over batch x rows x columns: no tile-shape waste on 20x20 maps) x 16*NT
//     channels; A = weight fragments from LDS (NT ds_read_b128 per k-step, shared by MT*NT MFMAs), B = pixel fragments by
//     range-checked buffer loads straight from L2 (padding taps / M tail = hardware zeros; a 3x3 stride-1 conv re-reads its input 9x
//     through the vector-memory path, 2.25x at stride 2 -- 236 MB per launch at most for these layers, < 7 us of the 64 B/clk path);
//   * K is walked in units of 2 k-steps (64 channels of one tap); a ring of NB unit buffers keeps NB - 1 units of pixel loads in
//     flight ahead of the MFMAs (NB = 3 for the big layers; NB = 9 -- a whole row of taps -- where a layer has so few wave tiles that
//     a wave's K walk is one dependent chain of memory round trips), and the ring does NOT drain between tiles: the last units of a
//     tile prefetch the first ones of the wave's next tile, so its loads fly during the epilogue;
//   * a workgroup owns a CONTIGUOUS run of wave tiles and its 8 waves take neighbouring tiles: the 9x (stride 1) / 2.25x (stride 2)
//     re-reads of an input line come from waves of the same CU / XCD (L1 / L2 hits instead of Infinity-Cache round trips);
//   * at most 256 VGPRs (2 waves per SIMD), grid = one workgroup per CU and channel tile.
template <int NT, int MT, int UPT, int S, int NB>
__global__ __launch_bounds__(512, 2) void conv3s_kernel(ConvP p) {
  typedef f16 T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* wl = reinterpret_cast<T*>(smem);
  constexpr int BN = 16 * NT, CIN = 64 * UPT, NU = 9 * UPT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n_base = blockIdx.y * BN;
  {  // weight tile: one contiguous block (LDS row stride == packed row stride)
    const T* wg = (const T*)p.w + (long)n_base * p.Kpad;
    const int nvec = BN * (p.Kpad >> 3);
    for (int v0 = threadIdx.x; v0 < nvec; v0 += 512 * 8) {
      Vec8<T> w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v0 + u * 512 < nvec) w[u].load(wg + (long)(v0 + u * 512) * 8);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v0 + u * 512 < nvec) w[u].store(wl + (v0 + u * 512) * 8);
    }
  }
  __syncthreads();
  const long M = (long)p.B * p.Ho * p.Wo;
  const int hw = p.Ho * p.Wo;
  const T* wlane = wl + r * p.LSw + 8 * g;
  const int BNp = 16 * p.NTpack;
  const int ch0 = (n_base / BNp) * BNp + g * 4 * p.NTpack + 4 * ((n_base % BNp) >> 4);
  const __amdgpu_buffer_rsrc_t rs = ey_rsrc(p.src[0], p.srcBytes[0]);
  const int Cs = p.srcCs[0], rowB = p.W * Cs * 2;  // bytes per input row

  // per tile and pixel block: byte offset of the window origin (tap 0,0; may be "negative" = wraps, only used when the tap is valid)
  // and the 9-bit mask of taps that fall inside the image
  struct Desc { int base[MT]; unsigned vm[MT]; };
  auto setup = [&](long tile, Desc& d) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const long m = tile * (16 * MT) + mt * 16 + r;
      const bool pv = m < M;
      const long mm = pv ? m : 0;
      const int b = (int)(mm / hw), rem = (int)(mm - (long)b * hw);
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int iy0 = oy * S - 1, ix0 = ox * S - 1;
      d.base[mt] = ((b * p.H + iy0) * p.W + ix0) * Cs * 2 + 16 * g;
      unsigned vm = 0u;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = iy0 + t / 3, ix = ix0 + t % 3;
        vm |= (pv && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (1u << t) : 0u;
      }
      d.vm[mt] = vm;
    }
  };
  Vec8<T> bq[NB][MT][2];
  auto issue = [&](const Desc& d, int u, Vec8<T> (&bf)[MT][2]) {
    const int tap = u / UPT, c0 = (u - tap * UPT) * 64;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int toff = ky * rowB + (kx * Cs + c0) * 2;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned vo = ((d.vm[mt] >> tap) & 1u) ? (unsigned)(d.base[mt] + toff) : EY_OOB;
      BufLoad8<T>::load(bf[mt][0], rs, vo, 0);
      BufLoad8<T>::load(bf[mt][1], rs, vo, 64);
    }
  };
  f32x4 acc[MT][NT];
  auto compute = [&](int u, const Vec8<T> (&bf)[MT][2]) {
    const T* wk = wlane + u * 64;  // k = tap * CIN + c0 = u * 64
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Vec8<T> af[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) af[nt].load(wk + nt * 16 * p.LSw + ks * 32);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16(af[nt], bf[mt][ks], acc[mt][nt]);
    }
  };

  // workgroup b owns tiles [b * tpw, (b + 1) * tpw); wave w takes b * tpw + w, + 8, ...
  const long lb = p.xcd ? ey_xcd_block(blockIdx.x, gridDim.x) : blockIdx.x;  // the runs of one XCD's workgroups are contiguous: shared window rows in one L2
  const long tpw = (p.ntile + gridDim.x - 1) / gridDim.x, tend = min(p.ntile, (lb + 1) * tpw);
  long tile = lb * tpw + wave;
  if (tile >= tend) return;
  Desc cur, nxt;
  setup(tile, cur);
#pragma unroll
  for (int j = 0; j < NB - 1; ++j) issue(cur, j, bq[j]);
  while (true) {
    const long ntl = tile + 8;
    const bool more = ntl < tend;
    if (more) setup(ntl, nxt);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4)0.f;
#pragma unroll 1
    for (int u = 0; u < NU; u += NB) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int uu = u + j + NB - 1;
        if (uu < NU) issue(cur, uu, bq[(j + NB - 1) % NB]);
        else if (more) issue(nxt, uu - NU, bq[(j + NB - 1) % NB]);  // the next tile's first units: in flight during this tile's epilogue
        compute(u + j, bq[j]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const long m = tile * (16 * MT) + mt * 16 + r;
      if (m < M) {
        const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
        const int oy = rem / p.Wo;
        conv_epilogue<T, NT>(p, acc[mt], m, b, oy, rem - oy * p.Wo, ch0, 0);
      }
    }
    if (!more) break;
    cur = nxt;
    tile = ntl;
  }
}

// ================================================================================================================
// Register-stationary 3x3 kernel for Cin == 16 (layer 1: 16->32 stride 2 at 320x320; the 16->8 sub-band convs): K = 144 is
// nine 16-channel taps, i.e. nine v_mfma_f32_16x16x16_f16 per 16 pixels per 16 output channels, and the whole weight tile is
// 9*NT two-register fragments -- it lives in registers.  A persistent wave walks 16-pixel row segments: 9 range-checked 8-byte
// buffer loads per lane (a pixel's 16 channels = the 4 lanes of its k-groups = one 32-byte segment; zero padding for free; the next
// segment's loads already in flight), 9*NT MFMAs, the shared epilogue.  No LDS, no barrier: these layers move 50-160 MB and were
// bound by staging / per-tile overheads (2.3 TB/s), not by arithmetic.
typedef f16 f16x4v __attribute__((ext_vector_type(4)));
template <int NT, int S>
__global__ __launch_bounds__(256) void conv3r_kernel(ConvP p) {
  typedef f16 T;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int grp = blockIdx.z;
  const __amdgpu_buffer_rsrc_t rs = ey_rsrc((const T*)p.src[0] + (long)grp * p.srcG, p.srcBytes[0]);
  f16x4v af[9][NT];
  {
    const T* wg = (const T*)p.w + (long)min(grp, p.wGmax) * p.wG;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) af[tap][nt] = *reinterpret_cast<const f16x4v*>(wg + (long)(nt * 16 + r) * p.Kpad + tap * 16 + 4 * g);
  }
  const int tilesX = (p.Wo + 15) >> 4;
  const int ntile = p.B * p.Ho * tilesX;  // < 2^31 (host check)
  const int nwave = gridDim.x * 4, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  // each wave owns a CONTIGUOUS run of row segments: (image, row, segment) advance incrementally, no division per tile
  const int per = (ntile + nwave - 1) / nwave;
  int tile = wid * per;
  const int tend = min(ntile, tile + per);
  if (tile >= tend) return;
  const int ch0 = g * 4 * NT;  // NT == NTpack, one channel tile
  int row0 = tile / tilesX;
  int ntx = tile - row0 * tilesX, nb = row0 / p.Ho, noy = row0 - nb * p.Ho;  // coordinates of the NEXT segment to issue
  auto issue = [&](f16x4v (&bq)[9], int& b, int& oy, int& ox) {
    b = nb; oy = noy; ox = ntx * 16 + r;
    const int ix0 = ox * S - 1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * S - 1 + ky;
      const bool yok = iy >= 0 && iy < p.H && ox < p.Wo;
      const int rowoff = ((b * p.H + iy) * p.W + ix0) * p.srcCs[0] + 4 * g;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ix0 + kx;
        const unsigned off = (yok && ix >= 0 && ix < p.W) ? (unsigned)((rowoff + kx * p.srcCs[0]) * 2) : EY_OOB;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
        bq[ky * 3 + kx] = __builtin_bit_cast(f16x4v, t);
      }
    }
    if (++ntx == tilesX) { ntx = 0; if (++noy == p.Ho) { noy = 0; ++nb; } }
  };
  f16x4v cur[9], nxt[9];
  int b = 0, oy = 0, ox = 0, b2 = 0, oy2 = 0, ox2 = 0;
  issue(cur, b, oy, ox);
  for (; tile < tend; ++tile) {
    if (tile + 1 < tend) issue(nxt, b2, oy2, ox2);
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4)0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(af[tap][nt], cur[tap], acc[nt], 0, 0, 0);
    if (ox < p.Wo) conv_epilogue<T, NT>(p, acc, ((long)b * p.Ho + oy) * p.Wo + ox, b, oy, ox, ch0, grp);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) cur[tap] = nxt[tap];
    b = b2; oy = oy2; ox = ox2;
  }
}

// ================================================================================================================
// N-split pointwise kernel for SMALL maps with MANY channels (1x1, stride 1, <= 2 sources, K = 128 ... 512, Cout a multiple of 128; f16):
// the 20x20 / 40x40 layers of the neck and C2PSA (12.8 k / 51.2 k pixels at batch 32) are a few hundred 16-pixel tiles against a
// 64-256 KB weight matrix.  The weight-stationary kernel stages a [128][K] tile per workgroup for a handful of tiles of work (staging
// = most of its 17-19 us); the lean kernel would have every wave re-read its weight rows per 16 pixels.  Here the roles are swapped:
// a 512-thread workgroup owns 64 pixels and a slab of 128 * NTW output channels; wave w owns 16 * NTW of those channels and keeps
// ITS rows of the weight matrix -- KS * NTW fragments, read once from L2, no other wave needs them -- in registers; the 64 pixels
// (all K channels, both sources, nearest-x2 sources resolved on the way) go to LDS once and every wave reads its B fragments there.
// One barrier, 4 * KS * NTW MFMAs per wave, k-steps in the order of conv_ws_kernel (bit-identical results).
template <int KS, int NTW>
__global__ __launch_bounds__(512) void conv_pwn_kernel(ConvP p) {
  typedef f16 T;
  constexpr int MB = 4, PX = 16 * MB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* s_px = reinterpret_cast<T*>(smem);  // [PX][KP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int KP = p.LSw, KV = KS * 4;  // LDS pixel pitch (elements); 16-byte vectors per pixel
  const int M = p.B * p.Ho * p.Wo, m0 = blockIdx.x * PX;
  // this wave's channels: block of 128 (NTpack = 8) and n-blocks inside it
  const int blk = NTW == 2 ? 2 * blockIdx.y + (wave >> 2) : blockIdx.y;
  const int nt0 = NTW == 2 ? 2 * (wave & 3) : wave;
  Vec8<T> af[KS][NTW];
  {
    const T* wr = (const T*)p.w + (long)(blk * 128 + nt0 * 16 + r) * p.Kpad + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) af[ks][nt].load(wr + (long)nt * 16 * p.Kpad + ks * 32);
  }
  {  // ---- pixels -> LDS (all requests of a thread first)
    const __amdgpu_buffer_rsrc_t rs0 = ey_rsrc(p.src[0], p.srcBytes[0]);
    const __amdgpu_buffer_rsrc_t rs1 = p.nsrc == 2 ? ey_rsrc(p.src[1], p.srcBytes[1]) : rs0;
    const int C0 = p.srcC[0], hw = p.Ho * p.Wo;
    const bool geo = p.srcUp[0] || (p.nsrc == 2 && p.srcUp[1]);
    constexpr int NV = PX * KS * 4, U = (NV + 511) / 512;
    Vec8<T> t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int v = tid + u * 512, px = v / KV, c = (v - px * KV) * 8, m = m0 + px;
      const bool second = c >= C0;
      const int s = second ? 1 : 0, cc = second ? c - C0 : c;
      unsigned off = EY_OOB;
      if (v < NV && m < M) {
        if (geo) {
          const int b = m / hw, rem = m - b * hw, oy = rem / p.Wo, ox = rem - oy * p.Wo, up = p.srcUp[s];
          off = (unsigned)(((((b * (p.H >> up)) + (oy >> up)) * (p.W >> up) + (ox >> up)) * p.srcCs[s] + cc) * 2);
        } else {
          off = (unsigned)((m * p.srcCs[s] + cc) * 2);
        }
      }
      BufLoad8<T>::load(t[u], second ? rs1 : rs0, off);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int v = tid + u * 512, px = v / KV, c = (v - px * KV) * 8;
      if (v < NV) t[u].store(s_px + px * KP + c);
    }
  }
  __syncthreads();
  f32x4 acc[MB][NTW];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mb][nt] = (f32x4)0.f;
  const T* bp = s_px + r * KP + 8 * g;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    Vec8<T> bq[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) bq[mb].load(bp + mb * 16 * KP + ks * 32);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[mb][nt] = mma16(af[ks][nt], bq[mb], acc[mb][nt]);
  }
  // ---- epilogue: lane (r, g) holds channels ch0 .. ch0 + 4 NTW of pixel m0 + 16 mb + r (NTpack = 8 row permutation)
  const int ch0 = blk * 128 + 32 * g + 4 * nt0;
  float bias[4 * NTW];
#pragma unroll
  for (int i = 0; i < 4 * NTW; ++i) bias[i] = p.bias ? p.bias[ch0 + i] : 0.f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = m0 + mb * 16 + r;
    if (m >= M) continue;
    float v[4 * NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * nt + j] = acc[mb][nt][j] + bias[4 * nt + j];
    if (p.act == EY_ACT_SILU) {
#pragma unroll
      for (int i = 0; i < 4 * NTW; ++i) v[i] = v[i] * ey_sigmoid(v[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4 * NTW; ++i) v[i] = ey_act(v[i], p.act) * p.out_scale;
    }
    T* yp = (T*)p.y + (long)m * p.yCs + ch0;
    if constexpr (NTW == 2) {
      Vec8<T> o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.set(j, v[j]);
      o.store(yp);
    } else {
      const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
      *reinterpret_cast<f16x4*>(yp) = o;
    }
  }
}

// ================================================================================================================
// Lean pointwise kernel (1x1, stride 1, <= 2 sources) for maps up to ~80x80.  Measured on MI355X (tools/micro): a lone
// wave retires about one instruction per 2.5 ns and a kernel launch costs 1.6 us, so for layers that move a few MB the
// run time is the per-wave instruction count plus the memory round trips, not bandwidth.  Hence: no LDS, no barrier,
// no 64-bit or runtime-divisor arithmetic on the common path (GEO=false: pixel index == flat output index), pixel AND
// weight fragments by buffer loads whose per-step address is a scalar offset (one instruction per fragment), every
// fragment of up to BATCH k-steps in flight before the first MFMA, many small wave tiles (16 pixels x 16*NT channels)
// so that every SIMD has a few waves to overlap.
template <typename T, int NT, int BATCH, bool TWO, bool GEO>
__global__ __launch_bounds__(256) void conv_pw_kernel(ConvP p) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int m_tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m_tile >= (int)p.ntile) return;  // whole wave
  const int n_base = blockIdx.y * (16 * NT);
  const int M = p.B * p.Ho * p.Wo;  // < 2^31 (host check)
  const int m = m_tile * 16 + r;
  const bool pv = m < M;
  int b = 0, oy = 0, ox = 0;
  unsigned voff0, voff1 = EY_OOB;
  if constexpr (GEO) {  // an upsampled source or the bilinear addz needs (b, y, x)
    const int hw = p.Ho * p.Wo, mm = pv ? m : 0;
    b = mm / hw;
    const int rem = mm - b * hw;
    oy = rem / p.Wo;
    ox = rem - oy * p.Wo;
    const int u0 = p.srcUp[0];
    voff0 = pv ? (unsigned)(((((b * (p.H >> u0)) + (oy >> u0)) * (p.W >> u0) + (ox >> u0)) * p.srcCs[0] + 8 * g) * (int)sizeof(T)) : EY_OOB;
    if constexpr (TWO) {
      const int u1 = p.srcUp[1];
      voff1 = pv ? (unsigned)(((((b * (p.H >> u1)) + (oy >> u1)) * (p.W >> u1) + (ox >> u1)) * p.srcCs[1] + 8 * g) * (int)sizeof(T)) : EY_OOB;
    }
  } else {
    voff0 = pv ? (unsigned)((m * p.srcCs[0] + 8 * g) * (int)sizeof(T)) : EY_OOB;
    if constexpr (TWO) voff1 = pv ? (unsigned)((m * p.srcCs[1] + 8 * g) * (int)sizeof(T)) : EY_OOB;
  }
  const __amdgpu_buffer_rsrc_t rs0 = ey_rsrc(p.src[0], p.srcBytes[0]);
  const __amdgpu_buffer_rsrc_t rs1 = TWO ? ey_rsrc(p.src[1], p.srcBytes[1]) : rs0;
  // this wave's 16*NT packed weight rows: lane (r, g) reads row nt*16 + r, k offset 8g (+ scalar step offset)
  const __amdgpu_buffer_rsrc_t rw = ey_rsrc((const T*)p.w + (long)n_base * p.Kpad, (unsigned)(16 * NT * p.Kpad * (int)sizeof(T)));
  const unsigned wvoff = (unsigned)((r * p.Kpad + 8 * g) * (int)sizeof(T));
  const int rowblk = 16 * p.Kpad * (int)sizeof(T);
  const int C0 = p.srcC[0], S0 = (C0 + 31) >> 5;
  const int C1 = TWO ? p.srcC[1] : 0;
  const int nsteps = S0 + (TWO ? ((C1 + 31) >> 5) : 0);

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4)0.f;
  for (int t0 = 0; t0 < nsteps; t0 += BATCH) {
    Vec8<T> bf[BATCH], af[BATCH][NT];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int t = t0 + u;
      if (t < nsteps) {  // wave-uniform
        const bool second = TWO && t >= S0;
        const int c = (second ? t - S0 : t) << 5;           // channel offset inside the source
        const int kofs = second ? C0 + c : c;               // K offset inside the packed row
        const bool cok = (c + 8 * g) < (second ? C1 : C0);  // channel tail: range check -> zeros
        if (second) BufLoad8<T>::load(bf[u], rs1, cok ? voff1 : EY_OOB, c * (int)sizeof(T));
        else BufLoad8<T>::load(bf[u], rs0, cok ? voff0 : EY_OOB, c * (int)sizeof(T));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) BufLoad8<T>::load(af[u][nt], rw, wvoff, nt * rowblk + kofs * (int)sizeof(T));
      }
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      if (t0 + u < nsteps) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mma16(af[u][nt], bf[u], acc[nt]);
      }
    }
  }
  if (!pv) return;
  const int BNp = 16 * p.NTpack;
  const int ch0 = (n_base / BNp) * BNp + g * 4 * p.NTpack + 4 * ((n_base % BNp) >> 4);
  conv_epilogue<T, NT>(p, acc, m, b, oy, ox, ch0, 0);
}

// ================================================================================================================
// Two chained pointwise convs as ONE launch (f16; maps up to ~50 k pixels): the enhancer tail conv of a DSC3K2_Wavelet block
// (bias + bilinear-resized pre-activation term + SiLU * tanh(gamma) + residual, conv_epilogue) followed by the stacked cv1|cv2 conv
// of the DSC3k behind it -- both read and write the same 16 pixels, so a wave runs them back to back: the first conv exactly as
// conv_pw_kernel (its result is stored: the block's concat and the bottleneck residuals need it), then its own 16 x Cmid outputs come
// back as the B fragments of the second conv (lane (r, g) wrote channels 4 NT g ..., reads channels 32 s + 8 g ...: a cross-lane move
// through the just-written lines, ordered by a workgroup-scope release / acquire pair = one s_waitcnt), second conv, plain epilogue.
// Same k-step order as the two conv_pw_kernel launches: bit-identical.
template <int NT, bool AG>
__global__ __launch_bounds__(256) void conv_pwc_kernel(ConvP p, ConvP q) {
  typedef f16 T;
  constexpr int BATCH = NT <= 4 ? 4 : 2;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int m_tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m_tile >= (int)p.ntile) return;  // whole wave
  const int M = p.B * p.Ho * p.Wo;
  const int m = m_tile * 16 + r;
  const bool pv = m < M;
  const int hw = p.Ho * p.Wo, mm = pv ? m : 0;
  const int b = mm / hw, rem = mm - b * hw, oy = rem / p.Wo, ox = rem - oy * p.Wo;
  const int ch0 = g * 4 * NT;  // one channel tile: NT == NTpack
  f32x4 acc[NT];
  {  // ---- first conv
    const unsigned voff = pv ? (unsigned)((m * p.srcCs[0] + 8 * g) * 2) : EY_OOB;
    const __amdgpu_buffer_rsrc_t rs = ey_rsrc(p.src[0], p.srcBytes[0]);
    const __amdgpu_buffer_rsrc_t rw = ey_rsrc(p.w, (unsigned)(16 * NT * p.Kpad * 2));
    const unsigned wvoff = (unsigned)((r * p.Kpad + 8 * g) * 2);
    const int rowblk = 16 * p.Kpad * 2, nsteps = p.srcC[0] >> 5;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4)0.f;
    for (int t0 = 0; t0 < nsteps; t0 += BATCH) {
      Vec8<T> bf[BATCH], af[BATCH][NT];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        if (t0 + u < nsteps) {
          const int c = (t0 + u) << 5;
          BufLoad8<T>::load(bf[u], rs, voff, c * 2);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) BufLoad8<T>::load(af[u][nt], rw, wvoff, nt * rowblk + c * 2);
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        if (t0 + u < nsteps) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = mma16(af[u][nt], bf[u], acc[nt]);
        }
      }
    }
    if (pv) conv_epilogue<T, NT>(p, acc, m, b, oy, ox, ch0, 0);
  }
  // the 16 x Cmid tile this wave just stored is its own input now, laid out across other lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the wave's stores are complete (write-through to L2; the CU's L1 sees its own CU's stores)
  if constexpr (AG) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // tune pwc = 2: additionally drop the CU's L1 (the by-the-book form for data another CU wrote; not needed here)
  else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  {  // ---- second conv
    const unsigned voff = pv ? (unsigned)((m * q.srcCs[0] + 8 * g) * 2) : EY_OOB;
    const __amdgpu_buffer_rsrc_t rs = ey_rsrc(q.src[0], q.srcBytes[0]);
    const __amdgpu_buffer_rsrc_t rw = ey_rsrc(q.w, (unsigned)(16 * NT * q.Kpad * 2));
    const unsigned wvoff = (unsigned)((r * q.Kpad + 8 * g) * 2);
    const int rowblk = 16 * q.Kpad * 2, nsteps = q.srcC[0] >> 5;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4)0.f;
    for (int t0 = 0; t0 < nsteps; t0 += BATCH) {
      Vec8<T> bf[BATCH], af[BATCH][NT];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        if (t0 + u < nsteps) {
          const int c = (t0 + u) << 5;
          BufLoad8<T>::load(bf[u], rs, voff, c * 2);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) BufLoad8<T>::load(af[u][nt], rw, wvoff, nt * rowblk + c * 2);
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        if (t0 + u < nsteps) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = mma16(af[u][nt], bf[u], acc[nt]);
        }
      }
    }
    if (pv) conv_epilogue<T, NT>(q, acc, m, b, oy, ox, ch0, 0);
  }
}

// ================================================================================================================
// Register-stationary pointwise kernel for LARGE maps with few channels (1x1, stride 1, Cin <= 128, <= 2 sources): the whole
// [16*NT][K] weight tile of a wave is KS*NT MFMA fragments -- it lives in registers for the life of a persistent wave, as
// does the bias.  A wave then walks 16-pixel tiles: KS buffer loads (next tile's already in flight), KS*NT MFMAs, a short
// epilogue, one or two wide stores -- a few dozen instructions per tile, no LDS, no barrier.  These layers (160x160 and 80x80
// maps, 16-96 channels) move 50-100 MB and are bound by how fast waves can issue loads/stores, not by arithmetic.
template <typename T, int NT, int KS, bool TWO, bool GEO>
__global__ __launch_bounds__(256) void conv_pwr_kernel(ConvP p) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int n_base = blockIdx.y * (16 * NT);
  const int M = p.B * p.Ho * p.Wo;  // < 2^31 (host check)
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
  const __amdgpu_buffer_rsrc_t rs0 = ey_rsrc(p.src[0], p.srcBytes[0]);
  const __amdgpu_buffer_rsrc_t rs1 = TWO ? ey_rsrc(p.src[1], p.srcBytes[1]) : rs0;
  const int C0 = p.srcC[0], S0 = (C0 + 31) >> 5, C1 = TWO ? p.srcC[1] : 0;
  // ---- weights and bias -> registers (once)
  Vec8<T> af[KS][NT];
  {
    const __amdgpu_buffer_rsrc_t rw = ey_rsrc((const T*)p.w + (long)n_base * p.Kpad, (unsigned)(16 * NT * p.Kpad * (int)sizeof(T)));
    const unsigned wvoff = (unsigned)((r * p.Kpad + 8 * g) * (int)sizeof(T));
    const int rowblk = 16 * p.Kpad * (int)sizeof(T);
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const bool second = TWO && t >= S0;
      const int kofs = second ? C0 + ((t - S0) << 5) : (t << 5);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) BufLoad8<T>::load(af[t][nt], rw, wvoff, nt * rowblk + kofs * (int)sizeof(T));
    }
  }
  const int BNp = 16 * p.NTpack;
  const int ch0 = (n_base / BNp) * BNp + g * 4 * p.NTpack + 4 * ((n_base % BNp) >> 4);
  float bias[4 * NT];
#pragma unroll
  for (int i = 0; i < 4 * NT; ++i) bias[i] = (p.bias && ch0 + i < p.Cout) ? p.bias[ch0 + i] : 0.f;
  // per-step channel-tail predicate and scalar byte offset inside the source
  bool cok[KS];
  int coff[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t) {
    const bool second = TWO && t >= S0;
    const int c = (second ? t - S0 : t) << 5;
    cok[t] = (c + 8 * g) < (second ? C1 : C0);
    coff[t] = c * (int)sizeof(T);
  }
  const int hw = p.Ho * p.Wo;
  const int u0 = p.srcUp[0], u1 = TWO ? p.srcUp[1] : 0;

  auto issue = [&](int tile, Vec8<T> (&bf)[KS], int& b, int& oy, int& ox) {
    const int m = tile * 16 + r;
    const bool pv = m < M;
    unsigned v0, v1 = EY_OOB;
    if constexpr (GEO) {
      const int mm = pv ? m : 0;
      b = mm / hw;
      const int rem = mm - b * hw;
      oy = rem / p.Wo;
      ox = rem - oy * p.Wo;
      v0 = pv ? (unsigned)(((((b * (p.H >> u0)) + (oy >> u0)) * (p.W >> u0) + (ox >> u0)) * p.srcCs[0] + 8 * g) * (int)sizeof(T)) : EY_OOB;
      if constexpr (TWO) v1 = pv ? (unsigned)(((((b * (p.H >> u1)) + (oy >> u1)) * (p.W >> u1) + (ox >> u1)) * p.srcCs[1] + 8 * g) * (int)sizeof(T)) : EY_OOB;
    } else {
      v0 = pv ? (unsigned)((m * p.srcCs[0] + 8 * g) * (int)sizeof(T)) : EY_OOB;
      if constexpr (TWO) v1 = pv ? (unsigned)((m * p.srcCs[1] + 8 * g) * (int)sizeof(T)) : EY_OOB;
    }
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      if (TWO && t >= S0) BufLoad8<T>::load(bf[t], rs1, cok[t] ? v1 : EY_OOB, coff[t]);
      else BufLoad8<T>::load(bf[t], rs0, cok[t] ? v0 : EY_OOB, coff[t]);
    }
  };

  Vec8<T> cur[KS], nxt[KS];
  int b = 0, oy = 0, ox = 0, nb = 0, noy = 0, nox = 0;
  int tile = wave_id;
  if (tile < (int)p.ntile) issue(tile, cur, b, oy, ox);
  for (; tile < (int)p.ntile; tile += nwave) {
    const int ntile = tile + nwave;
    if (ntile < (int)p.ntile) issue(ntile, nxt, nb, noy, nox);
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4)0.f;
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mma16(af[t][nt], cur[t], acc[nt]);
    const int m = tile * 16 + r;
    if (m < M) {
      float v[4 * NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * nt + j] = acc[nt][j] + bias[4 * nt + j];
      if constexpr (GEO) {
        if (p.addz) {  // bilinear (align_corners=False) resize of the half-resolution pre-activation term, see conv_epilogue
          const int Hz = p.Hz, Wz = p.Wz;
          const float sy = fmaxf(p.zsy * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(p.zsx * (ox + 0.5f) - 0.5f, 0.f);
          const int y0 = (int)sy, x0 = (int)sx;
          const int y1 = min(y0 + 1, Hz - 1), x1 = min(x0 + 1, Wz - 1);
          const float ly1 = sy - y0, lx1 = sx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
          const T* z = (const T*)p.addz + ch0;
          const T* z00 = z + ((long)(b * Hz + y0) * Wz + x0) * p.addzCs;
          const T* z01 = z + ((long)(b * Hz + y0) * Wz + x1) * p.addzCs;
          const T* z10 = z + ((long)(b * Hz + y1) * Wz + x0) * p.addzCs;
          const T* z11 = z + ((long)(b * Hz + y1) * Wz + x1) * p.addzCs;
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            float a00[4], a01[4], a10[4], a11[4];
            load4(z00 + 4 * q, a00); load4(z01 + 4 * q, a01); load4(z10 + 4 * q, a10); load4(z11 + 4 * q, a11);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * q + j] += ly0 * (lx0 * a00[j] + lx1 * a01[j]) + ly1 * (lx0 * a10[j] + lx1 * a11[j]);
          }
        }
      }
      if (p.act == EY_ACT_SILU && p.out_scale == 1.f) {
#pragma unroll
        for (int i = 0; i < 4 * NT; ++i) v[i] = v[i] * ey_sigmoid(v[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 4 * NT; ++i) v[i] = ey_act(v[i], p.act) * p.out_scale;
      }
      T* yp = (T*)p.y + (long)m * p.yCs + ch0;
      const T* rp = p.res ? (const T*)p.res + (long)m * p.resCs + ch0 : nullptr;
      if (sizeof(T) == 2 && NT % 2 == 0 && p.vec_store > 1) {
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {
          if (rp) {
            Vec8<T> rr;
            rr.load(rp + 8 * q);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[8 * q + j] += rr.get(j);
          }
          Vec8<T> o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o.set(j, v[8 * q + j]);
          o.store(yp + 8 * q);
        }
      } else {  // 4-element accesses (the host only dispatches here when the views allow them and Cout % (4*NT...) holds)
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          if (rp) {
            float rr[4];
            load4(rp + 4 * q, rr);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * q + j] += rr[j];
          }
          store4(yp + 4 * q, v + 4 * q);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < KS; ++t) cur[t] = nxt[t];
    b = nb; oy = noy; ox = nox;
  }
}

// ================================================================================================================
// Two chained pointwise convs in ONE kernel, registers only (f16; the last two 1x1 convs of the Detect class tower):
//     y = act2( W2 . act1( W1 . x + b1 ) + b2 )
// After the first GEMM lane (pixel r, group g) holds 4*NT1 consecutive mid channels of its pixel -- and a contraction may run over
// its index in ANY order as long as both operands agree.  So the second GEMM's k-slot (step s, 8g + j) is DEFINED as mid channel
// g*4*NT1 + 8s + j: the activated, f16-rounded values of a lane are its own B fragments, nothing moves between lanes, no LDS; W2 is
// packed by the ordinary ey_conv_pack_weight() after its input columns were permuted accordingly (ey_conv_chain_kperm).
// Persistent waves, both weight tiles and both biases in registers, next tile's pixels in flight (as conv_pwr_kernel).
struct ChainP {
  const void* w2; const float* b2; int act2, Cout2, Kpad2;
  void* y2; int y2Cs;
};
template <int NT1, int KS1, int NT2>
__global__ __launch_bounds__(256) void conv_pw2_kernel(ConvP p, ChainP q) {
  typedef f16 T;
  constexpr int KS2 = (4 * NT1 + 7) / 8;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int M = p.B * p.Ho * p.Wo;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
  const __amdgpu_buffer_rsrc_t rs0 = ey_rsrc(p.src[0], p.srcBytes[0]);
  const int C0 = p.srcC[0];
  Vec8<T> a1[KS1][NT1], a2[KS2][NT2];
  {
    const __amdgpu_buffer_rsrc_t rw = ey_rsrc(p.w, (unsigned)(16 * NT1 * p.Kpad * 2));
    const unsigned wvoff = (unsigned)((r * p.Kpad + 8 * g) * 2);
#pragma unroll
    for (int t = 0; t < KS1; ++t)
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) BufLoad8<T>::load(a1[t][nt], rw, wvoff, (nt * 16 * p.Kpad + t * 32) * 2);
    const __amdgpu_buffer_rsrc_t rw2 = ey_rsrc(q.w2, (unsigned)(16 * NT2 * q.Kpad2 * 2));
    const unsigned wvoff2 = (unsigned)((r * q.Kpad2 + 8 * g) * 2);
#pragma unroll
    for (int t = 0; t < KS2; ++t)
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) BufLoad8<T>::load(a2[t][nt], rw2, wvoff2, (nt * 16 * q.Kpad2 + t * 32) * 2);
  }
  const int ch1 = g * 4 * NT1, ch2 = g * 4 * NT2;
  float b1[4 * NT1], b2[4 * NT2];
#pragma unroll
  for (int i = 0; i < 4 * NT1; ++i) b1[i] = (p.bias && ch1 + i < p.Cout) ? p.bias[ch1 + i] : 0.f;
#pragma unroll
  for (int i = 0; i < 4 * NT2; ++i) b2[i] = (q.b2 && ch2 + i < q.Cout2) ? q.b2[ch2 + i] : 0.f;
  bool cok[KS1];
#pragma unroll
  for (int t = 0; t < KS1; ++t) cok[t] = (32 * t + 8 * g) < C0;
  auto issue = [&](int tile, Vec8<T> (&bf)[KS1]) {
    const int m = tile * 16 + r;
    const unsigned v0 = m < M ? (unsigned)((m * p.srcCs[0] + 8 * g) * 2) : EY_OOB;
#pragma unroll
    for (int t = 0; t < KS1; ++t) BufLoad8<T>::load(bf[t], rs0, cok[t] ? v0 : EY_OOB, t * 64);
  };
  Vec8<T> cur[KS1], nxt[KS1];
  int tile = wave_id;
  if (tile < (int)p.ntile) issue(tile, cur);
  for (; tile < (int)p.ntile; tile += nwave) {
    if (tile + nwave < (int)p.ntile) issue(tile + nwave, nxt);
    f32x4 acc[NT1];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) acc[nt] = (f32x4)0.f;
#pragma unroll
    for (int t = 0; t < KS1; ++t)
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) acc[nt] = mma16(a1[t][nt], cur[t], acc[nt]);
    // mid activations (rounded to f16 like the tensor the unfused form writes) = B fragments of the second GEMM
    Vec8<T> mid[KS2];
    {
      float mv[8 * KS2];
#pragma unroll
      for (int idx = 0; idx < 8 * KS2; ++idx) mv[idx] = idx < 4 * NT1 ? acc[(idx >> 2) < NT1 ? (idx >> 2) : 0][idx & 3] + b1[idx < 4 * NT1 ? idx : 0] : 0.f;
      ey_act_n(mv, p.act);
#pragma unroll
      for (int t = 0; t < KS2; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int idx = 8 * t + j;
          mid[t].set(j, (idx < 4 * NT1 && ch1 + idx < p.Cout) ? mv[idx] : 0.f);
        }
    }
    f32x4 acc2[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) acc2[nt] = (f32x4)0.f;
#pragma unroll
    for (int t = 0; t < KS2; ++t)
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) acc2[nt] = mma16(a2[t][nt], mid[t], acc2[nt]);
    const int m = tile * 16 + r;
    if (m < M) {
      T* yp = (T*)q.y2 + (long)m * q.y2Cs + ch2;
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) {
        if (ch2 + 4 * nt + 4 <= q.Cout2) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc2[nt][j] + b2[4 * nt + j];
          ey_act_n(v, q.act2);
          store4(yp + 4 * nt, v);
        } else {  // channel tail (Cout not a multiple of 4, e.g. nc = 10)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (ch2 + 4 * nt + j < q.Cout2) yp[4 * nt + j] = (T)ey_act(acc2[nt][j] + b2[4 * nt + j], q.act2);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < KS1; ++t) cur[t] = nxt[t];
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int conv_nt(int Cout) {  // channels per block tile / 16
  if (Cout <= 16) return 1;
  if (Cout <= 32) return 2;
  if (Cout <= 64) return 4;
  if (Cout <= 80) return 5;
  if (Cout <= 128) return 8;
  if (Cout % 128 == 0) return 8;
  if (Cout % 80 == 0) return 5;
  if (Cout % 64 == 0) return 4;
  return 8;
}
static int conv_cout_pad(int Cout) { int bn = 16 * conv_nt(Cout); return (Cout + bn - 1) / bn * bn; }
static int conv_kpad(int Cin, int k, int es) { return ey_conv_kpad(k * k * Cin, es); }

#if EY_CONV_PART == 16
extern "C" size_t ey_conv_packed_bytes(int dtype, int Cout, int Cin, int k) {
  return (size_t)conv_cout_pad(Cout) * conv_kpad(Cin, k, dtype == EY_F16 ? 2 : 4) * (dtype == EY_F16 ? 2 : 4);
}

extern "C" int ey_conv_pack_weight(int dtype, int Cout, int Cin, int k, const float* w, void* out, size_t out_bytes) {
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "pack: bad dtype %d", dtype);
  EY_CHECK(Cout > 0 && Cin > 0 && (k == 1 || k == 3), "pack: Cout=%d Cin=%d k=%d", Cout, Cin, k);
  EY_CHECK(out_bytes >= ey_conv_packed_bytes(dtype, Cout, Cin, k), "pack: output buffer too small");
  const int NT = conv_nt(Cout), BN = 16 * NT, Kp = conv_kpad(Cin, k, dtype == EY_F16 ? 2 : 4), rows = conv_cout_pad(Cout);
  for (int row = 0; row < rows; ++row) {
    // MFMA row rho = 4g+j of n-block nt inside block tile nb  <->  channel nb*BN + g*4NT + 4nt + j
    const int nb = row / BN, within = row % BN, nt = within / 16, rho = within % 16, g = rho / 4, j = rho % 4;
    const int ch = nb * BN + g * 4 * NT + 4 * nt + j;
    for (int kk = 0; kk < Kp; ++kk) {
      float val = 0.f;
      if (ch < Cout && kk < k * k * Cin) {
        const int tap = kk / Cin, c = kk % Cin, ky = tap / k, kx = tap % k;
        val = w[(((long)ch * Cin + c) * k + ky) * k + kx];
      }
      const long o = (long)row * Kp + kk;
      if (dtype == EY_F16) ((f16*)out)[o] = (f16)val;
      else ((float*)out)[o] = val;
    }
  }
  return EY_OK;
}

#endif

template <typename T, int NT, int MT>
static bool conv_lds_ok() {  // one-time opt-in to > 64 KiB of dynamic LDS for the big-tile / f32 variants
  static const bool ok = [] {
    const size_t lds = 2 * (size_t)(16 * NT) * CONV_LS * sizeof(T);
    return lds <= 64 * 1024 ||
           hipFuncSetAttribute((const void*)conv_igemm_kernel<T, NT, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  }();
  return ok;
}

template <typename T, int NT>
static bool launch_conv(const ConvP& p, int ngroup, hipStream_t st) {
  const long M = (long)p.B * p.Ho * p.Wo;
  const int ntiles = (p.Cout + 16 * NT - 1) / (16 * NT);
  const long blocks2 = (M + 127) / 128 * ntiles * ngroup;
  const size_t lds = 2 * (size_t)(16 * NT) * CONV_LS * sizeof(T);
  if (blocks2 >= 512) {
    if (!conv_lds_ok<T, NT, 2>()) return false;
    dim3 grid((unsigned)((M + 127) / 128), ntiles, ngroup);
    hipLaunchKernelGGL((conv_igemm_kernel<T, NT, 2>), grid, dim3(256), lds, st, p);
  } else {
    if (!conv_lds_ok<T, NT, 1>()) return false;
    dim3 grid((unsigned)((M + 63) / 64), ntiles, ngroup);
    hipLaunchKernelGGL((conv_igemm_kernel<T, NT, 1>), grid, dim3(256), lds, st, p);
  }
  return true;
}

// ---- tunables (defaults measured on MI355X; EY_* environment variables override them for sweeps)
#include <stdlib.h>
#include "tune.h"
// kind*1000 + NT*10 + x of the kernel the last ey_conv2d launched (profiling labels); defined in the f16 translation unit
#if EY_CONV_PART == 16
thread_local int g_last_variant = 0;
#else
extern thread_local int g_last_variant;
#endif
// ---- weight-stationary dispatch
static int ws_ls(int Kpad) { return Kpad; }  // conv_kpad() already makes the row pitch conflict-free for the LDS fragment reads
static const int WS_NT[5] = {8, 5, 4, 2, 1};
// largest NT (<= the packing NT, dividing it into whole 16-row blocks) whose weight tile fits `budget` bytes of LDS
static int ws_pick_nt(int Cout, int Kpad, int es, size_t budget) {
  const int ntp = conv_nt(Cout);
  for (int i = 0; i < 5; ++i) {
    const int nt = WS_NT[i];
    if (nt > ntp || ntp % nt) continue;
    if ((size_t)16 * nt * ws_ls(Kpad) * es <= budget) return nt;
  }
  return 0;
}

template <typename T, int NT, int MT, int KS>
static bool ws_launch(ConvP p, int ngroup, hipStream_t st) {
  const size_t lds = (size_t)16 * NT * p.LSw * sizeof(T);
  static size_t reserved = 0;
  if (lds > 64 * 1024 && lds > reserved) {
    if (hipFuncSetAttribute((const void*)conv_ws_kernel<T, NT, MT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    reserved = lds;
  }
  const long M = (long)p.B * p.Ho * p.Wo;
  p.ntile = (M + 16 * MT - 1) / (16 * MT);
  const int ntiles_n = (conv_cout_pad(p.Cout)) / (16 * NT);
  // resident workgroups per CU: LDS AND registers decide (a 512-thread workgroup of a 172-VGPR instantiation fits once per
  // CU whatever its LDS footprint); a persistent grid larger than that runs in two rounds and stages every weight tile twice
  static size_t occ_lds = ~(size_t)0;
  static int occ = 1;
  if (occ_lds != lds) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv_ws_kernel<T, NT, MT, KS>, 512, lds) != hipSuccess || n < 1) n = 1;
    occ = n;
    occ_lds = lds;
  }
  const int wg_per_cu = occ < (int)tune().ws_wg_cu ? occ : (int)tune().ws_wg_cu;
  // persistent grid: one workgroup per resident slot; tiles are dealt round-robin over workgroups first, then waves,
  // so a small layer still spreads over all CUs
  long cap = (long)256 * wg_per_cu / ((long)ntiles_n * ngroup);
  if (cap < 1) cap = 1;
  cap = cap / tune().grid_div > 0 ? cap / tune().grid_div : 1;
  long gx = p.ntile < cap ? p.ntile : cap;
  if (tune().tiles_per_wave > 0) {
    long want = (p.ntile + 8 * tune().tiles_per_wave - 1) / (8 * tune().tiles_per_wave);
    if (want < 1) want = 1;
    if (want < gx) gx = want;
  }
  dim3 grid((unsigned)gx, ntiles_n, ngroup);
  hipLaunchKernelGGL((conv_ws_kernel<T, NT, MT, KS>), grid, dim3(512), lds, st, p);
  return true;
}

template <typename T, int NT, int KS>
static bool ws_launch_mt(const ConvP& p, int ngroup, hipStream_t st) {
  const long M = (long)p.B * p.Ho * p.Wo;
  // enough wave tiles to give every SIMD work: 2 pixel blocks per wave when M is large, else 1
  if (M >= tune().mt2_min_m) return ws_launch<T, NT, 2, KS>(p, ngroup, st);  // measured: below this, more (smaller) wave tiles hide latency better
  return ws_launch<T, NT, 1, KS>(p, ngroup, st);
}

template <typename T, int KS>
static bool ws_launch_nt(const ConvP& p, int nt, int ngroup, hipStream_t st) {
  switch (nt) {
    case 1: return ws_launch_mt<T, 1, KS>(p, ngroup, st);
    case 2: return ws_launch_mt<T, 2, KS>(p, ngroup, st);
    case 4: return ws_launch_mt<T, 4, KS>(p, ngroup, st);
    case 5: return ws_launch_mt<T, 5, KS>(p, ngroup, st);
    default: return ws_launch_mt<T, 8, KS>(p, ngroup, st);
  }
}

// ---- 3x3 halo-tile dispatch
template <typename T, int NT, int S>
static int halo_launch(ConvP p, int ngroup, hipStream_t st) {
  constexpr int MT = (S == 1) ? 2 : 1, TR = 8, TC = 16 * MT, HR = (TR - 1) * S + 3, HC = (TC - 1) * S + 3;
  const int C = p.srcC[0];
  const size_t lds = ((size_t)16 * NT * p.LSw + (size_t)HR * HC * (C + 8)) * sizeof(T);
  static size_t reserved = 0;
  if (lds > 64 * 1024 && lds > reserved) {
    if (hipFuncSetAttribute((const void*)conv3_halo_kernel<T, NT, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "conv: cannot reserve %zu B of LDS for the halo tile", lds);
    reserved = lds;
  }
  const long ntile = (long)p.B * ((p.Wo + TC - 1) / TC) * ((p.Ho + TR - 1) / TR);
  const int ntn = conv_cout_pad(p.Cout) / (16 * NT);
  static size_t occ_lds = ~(size_t)0;
  static int occ = 1;
  if (occ_lds != lds) {  // resident workgroups per CU from LDS and registers (see ws_launch)
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv3_halo_kernel<T, NT, S>, 512, lds) != hipSuccess || n < 1) n = 1;
    occ = n;
    occ_lds = lds;
  }
  const int per_cu = occ < 2 ? occ : 2;
  long gx = (long)256 * per_cu / ((long)ntn * ngroup);
  if (gx < 1) gx = 1;
  if (gx > ntile) gx = ntile;
  hipLaunchKernelGGL((conv3_halo_kernel<T, NT, S>), dim3((unsigned)gx, ntn, ngroup), dim3(512), lds, st, p);
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_conv2d(halo): %s", hipGetErrorString(e_));
  return 1;
}

// largest NT dividing the packing NT such that weights + halo fit LDS; 0 = does not fit
template <typename T>
static int halo_pick_nt(const ConvP& p, int S) {
  const int MT = (S == 1) ? 2 : 1, TC = 16 * MT, HR = 7 * S + 3, HC = (TC - 1) * S + 3, C = p.srcC[0];
  const size_t halo = (size_t)HR * HC * (C + 8) * sizeof(T);
  const int ntp = conv_nt(p.Cout);
  for (int i = 0; i < 5; ++i) {
    const int nt = WS_NT[i];
    if (nt > ntp || ntp % nt) continue;
    if (halo + (size_t)16 * nt * ws_ls(p.Kpad) * sizeof(T) <= 158 * 1024) return nt;
  }
  return 0;
}

template <typename T>
static int dispatch_halo(ConvP p, int ngroup, hipStream_t st) {
  if (p.k != 3 || p.nsrc != 1 || p.srcUp[0] || p.srcC[0] > 64 || p.srcC[0] < tune().halo_min_c) return 0;  // measured: wins for Cin=64 on large maps
  const int S = p.stride;
  {  // the per-thread register halo holds HV=10 vectors
    const int MT = (S == 1) ? 2 : 1, HR = 7 * S + 3, HC = (16 * MT - 1) * S + 3;
    if ((long)HR * HC * (p.srcC[0] >> 3) > 512L * 10) return 0;
  }
  const long npix = (long)p.B * p.H * p.W;
  const long bytes = ((npix - 1) * p.srcCs[0] + p.srcC[0]) * (long)sizeof(T);
  if (bytes >= (1L << 31) || p.srcG * (long)sizeof(T) * (ngroup - 1) >= (1L << 31)) return 0;
  p.srcBytes[0] = (unsigned)bytes;
  p.NTpack = conv_nt(p.Cout);
  p.LSw = ws_ls(p.Kpad);
  const int nt = halo_pick_nt<T>(p, S);
  if (!nt) return 0;
#define HALO(NTV)                                                     \
  case NTV: return S == 1 ? halo_launch<T, NTV, 1>(p, ngroup, st) : halo_launch<T, NTV, 2>(p, ngroup, st);
  switch (nt) {
    HALO(1) HALO(2) HALO(4) HALO(5) HALO(8)
  }
#undef HALO
  return 0;
}

// ---- 3x3 tile dispatch
template <typename T, int NT, int S>
static int tile_launch(ConvP p, int ngroup, hipStream_t st) {
  constexpr int TR = 8, TC = (S == 1) ? 32 : 16, HR = (TR - 1) * S + 3, HC = (TC - 1) * S + 3, LROW = (S == 1) ? HC : 2 * ((HC + 1) / 2);
  const bool wlds = tune().tile_wlds == 1 || (tune().tile_wlds == 2 && S == 1);
  const size_t lds = ((size_t)HR * LROW + (wlds ? 9 * 16 * NT : 0)) * 40 * sizeof(T);
  int tr = TR, tc = TC;
  if (S == 1 && tune().tile_flat) {  // flattened tile: the (rows x cols) with <= 256 pixels and <= 340 halo pixels that wastes the fewest slots
    double best = 0.0;
    const int cands[8] = {16, 20, 24, 28, 32, 36, 40, p.Wo};
    for (int i = 0; i < 8; ++i) {
      const int c = cands[i];
      if (c < 8 || c > 80) continue;
      int rr = 256 / c;
      while (rr > 1 && (rr + 2) * (c + 2) > 340) --rr;
      if (rr < 1 || (rr + 2) * (c + 2) > 340) continue;
      const long cov = (long)((p.Wo + c - 1) / c) * ((p.Ho + rr - 1) / rr) * 256;
      const double eff = (double)p.Wo * p.Ho / (double)cov;
      if (eff > best + 1e-9) { best = eff; tr = rr; tc = c; }
    }
  }
  p.tTR = tr; p.tTC = tc;
  const long tiles = (long)p.B * ((p.Wo + tc - 1) / tc) * ((p.Ho + tr - 1) / tr);
  const dim3 grid((unsigned)tiles, (unsigned)(conv_cout_pad(p.Cout) / (16 * NT)), (unsigned)ngroup);
  p.xcd = (int)((tune().xcd_map >> 4) & 1) && grid.y == 1 && grid.z == 1;  // (x alone decides the XCD only for a 1-D grid)
  static bool attr = false;
  if (!attr) {  // up to 46 + 46 KB of dynamic LDS
    (void)hipFuncSetAttribute((const void*)conv3_tile_kernel<T, NT, S, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    attr = true;
  }
  if (wlds) hipLaunchKernelGGL((conv3_tile_kernel<T, NT, S, true>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((conv3_tile_kernel<T, NT, S, false>), grid, dim3(256), lds, st, p);
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_conv2d(tile): %s", hipGetErrorString(e_));
  g_last_variant = 6000 + NT * 10 + S;
  return 1;
}

template <typename T>
static int dispatch_tile(ConvP p, int ngroup, hipStream_t st) {
  if (sizeof(T) != 2) return 0;  // f16 throughput mode (the f32 parity mode keeps the exact-f32 kernels below)
  if (p.k != 3 || p.nsrc != 1 || p.srcUp[0] || 9L * p.srcC[0] < tune().tile_mink) return 0;
  if (p.stride == 2 && (p.srcC[0] < tune().tile_s2_minc || (long)p.B * p.Ho * p.Wo < tune().tile_s2_minm)) return 0;
  const int ntp = conv_nt(p.Cout);
  int nt = ntp % 4 == 0 ? 4 : ntp == 2 ? 2 : ntp == 1 ? 1 : 0;
  if (!nt) return 0;
  // small maps: too few tiles to fill 256 CUs -> narrower channel tiles (more workgroups) beat the bigger register tile
  if (p.stride == 1 && tune().tile_minwg > 0) {
    const long tiles = (long)p.B * (((long)p.Ho * p.Wo + 239) / 240);
    while (nt > 1 && tiles * (conv_cout_pad(p.Cout) / (16 * nt)) * ngroup < tune().tile_minwg) nt >>= 1;
  }
  const long npix = (long)p.B * p.H * p.W;
  const long bytes = ((npix - 1) * p.srcCs[0] + p.srcC[0]) * (long)sizeof(T);
  if (bytes >= (1L << 31) || p.srcG * (long)sizeof(T) * (ngroup - 1) >= (1L << 31)) return 0;
  if ((long)conv_cout_pad(p.Cout) * p.Kpad * (long)sizeof(T) >= (1L << 31)) return 0;
  p.srcBytes[0] = (unsigned)bytes;
  p.NTpack = ntp;
  if constexpr (sizeof(T) == 2) {
    if (p.stride == 1) return nt == 4 ? tile_launch<T, 4, 1>(p, ngroup, st) : nt == 2 ? tile_launch<T, 2, 1>(p, ngroup, st) : tile_launch<T, 1, 1>(p, ngroup, st);
    return nt == 4 ? tile_launch<T, 4, 2>(p, ngroup, st) : nt == 2 ? tile_launch<T, 2, 2>(p, ngroup, st) : tile_launch<T, 1, 2>(p, ngroup, st);
  }
  return 0;
}

// ---- register-stationary 3x3 dispatch (Cin == 16)
template <int NT, int S>
static int c3r_launch(const ConvP& p, int ngroup, hipStream_t st) {
  static int occ = 0;
  if (!occ) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv3r_kernel<NT, S>, 256, 0) != hipSuccess || n < 1) n = 1;
    occ = n > 4 ? 4 : n;
  }
  const long ntile = (long)p.B * p.Ho * ((p.Wo + 15) / 16);
  if (ntile >= (1L << 31)) return 0;
  long gx = (long)256 * occ / ngroup / tune().grid_div;
  if (gx > (ntile + 3) / 4) gx = (ntile + 3) / 4;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((conv3r_kernel<NT, S>), dim3((unsigned)gx, 1, (unsigned)ngroup), dim3(256), 0, st, p);
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_conv2d(c3r): %s", hipGetErrorString(e_));
  g_last_variant = 7000 + NT * 10 + S;
  return 1;
}

template <typename T>
static int dispatch_c3r(ConvP p, int ngroup, hipStream_t st) {
  if (sizeof(T) != 2 || !tune().c3r) return 0;
  if (p.k != 3 || p.nsrc != 1 || p.srcUp[0] || p.srcC[0] != 16 || (p.srcCs[0] * 2) % 8) return 0;
  if (p.stride != 2 && tune().c3r < 2) return 0;  // measured: wins for the stride-2 layer (68 -> 58 us), loses 10 % to the tile kernel at stride 1
  const int ntp = conv_nt(p.Cout);
  if (ntp > 2 || conv_cout_pad(p.Cout) != 16 * ntp) return 0;
  const long npix = (long)p.B * p.H * p.W;
  const long bytes = ((npix - 1) * p.srcCs[0] + p.srcC[0]) * 2L;
  if (bytes >= (1L << 31) || p.srcG * 2L * (ngroup - 1) >= (1L << 31)) return 0;
  p.srcBytes[0] = (unsigned)bytes;
  p.NTpack = ntp;
  if constexpr (sizeof(T) == 2) {
    if (p.stride == 1) return ntp == 1 ? c3r_launch<1, 1>(p, ngroup, st) : c3r_launch<2, 1>(p, ngroup, st);
    return ntp == 1 ? c3r_launch<1, 2>(p, ngroup, st) : c3r_launch<2, 2>(p, ngroup, st);
  }
  return 0;
}

// ---- persistent 3x3 tile kernel dispatch (f16, Cin = 64, stride 1, Cout a multiple of 64 or exactly 32 / 16)
template <int NT>
static int c3p_launch(ConvP p, hipStream_t st) {
  // flattened tile: the (rows x cols) with <= 256 pixels and <= 340 halo pixels that wastes the fewest slots (as tile_launch)
  int tr = 8, tc = 32;
  {
    double best = 0.0;
    const int cands[8] = {16, 20, 24, 28, 32, 36, 40, p.Wo};
    for (int i = 0; i < 8; ++i) {
      const int c = cands[i];
      if (c < 8 || c > 80) continue;
      int rr = 256 / c;
      while (rr > 1 && (rr + 2) * (c + 2) > 340) --rr;
      if (rr < 1 || (rr + 2) * (c + 2) > 340) continue;
      const long cov = (long)((p.Wo + c - 1) / c) * ((p.Ho + rr - 1) / rr) * 256;
      const double eff = (double)p.Wo * p.Ho / (double)cov;
      if (eff > best + 1e-9) { best = eff; tr = rr; tc = c; }
    }
  }
  p.tTR = tr; p.tTC = tc;
  const long tiles = (long)p.B * ((p.Wo + tc - 1) / tc) * ((p.Ho + tr - 1) / tr);
  if (tiles >= (1L << 30)) return 0;
  const int ny = conv_cout_pad(p.Cout) / (16 * NT);
  const size_t lds = ((size_t)9 * 16 * NT + 340) * 80 * 2;
  static bool reserved = false;
  if (!reserved) {
    if (hipFuncSetAttribute((const void*)conv3p_kernel<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv3p_kernel<NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return 0;
    reserved = true;
  }
  long gx = 256 / ny;
  if (gx < 1) gx = 1;
  if (gx > tiles) gx = tiles;
  // the interleaved epilogue: bias + SiLU only, whole 16-byte-aligned channel tiles, output view addressable with 32-bit offsets
  const long ybytes = (((long)p.B * p.Ho * p.Wo - 1) * p.yCs + p.Cout) * 2L;
  const bool fast = tune().c3p_fast && p.bias && p.act == EY_ACT_SILU && p.out_scale == 1.f && !p.res && !p.addz && p.vec_store == 2 && p.Cout % (16 * NT) == 0 &&
                    ybytes < (1L << 31);
  p.srcBytes[1] = fast ? (unsigned)ybytes : 0u;
  const dim3 gg((unsigned)gx, (unsigned)ny, 1);
  if (fast) hipLaunchKernelGGL((conv3p_kernel<NT, true>), gg, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((conv3p_kernel<NT, false>), gg, dim3(256), lds, st, p);
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_conv2d(c3p): %s", hipGetErrorString(e_));
  g_last_variant = 9000 + NT * 10 + (fast ? 1 : 0);
  return 1;
}
template <typename T>
static int dispatch_c3p(ConvP p, int ngroup, hipStream_t st) {
  if constexpr (sizeof(T) != 2) return 0;
  else {
    if (!tune().c3p || p.k != 3 || p.stride != 1 || p.nsrc != 1 || p.srcUp[0] || ngroup != 1 || p.srcC[0] != 64) return 0;
    const int ntp = conv_nt(p.Cout);
    if (ntp % 4 != 0 || conv_cout_pad(p.Cout) % 64) return 0;
    const long M = (long)p.B * p.Ho * p.Wo;
    if (tune().c3p < 2 && M < tune().c3p_min_m) return 0;
    const long npix = (long)p.B * p.H * p.W;
    const long bytes = ((npix - 1) * p.srcCs[0] + p.srcC[0]) * 2L;
    if (bytes >= (1L << 31) || (long)conv_cout_pad(p.Cout) * p.Kpad * 2L >= (1L << 31)) return 0;
    p.srcBytes[0] = (unsigned)bytes;
    p.NTpack = ntp;
    return c3p_launch<4>(p, st);
  }
}

// ---- 3x3 stream kernel dispatch (f16, Cin in {64, 128, 256}, one source, no groups)
template <int NT, int MT, int UPT, int S, int NB>
static int c3s_launch2(ConvP p, hipStream_t st) {
  const size_t lds = (size_t)16 * NT * p.LSw * 2;
  static size_t reserved = 0;
  if (lds > 64 * 1024 && lds > reserved) {
    if (hipFuncSetAttribute((const void*)conv3s_kernel<NT, MT, UPT, S, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    reserved = lds;
  }
  const long M = (long)p.B * p.Ho * p.Wo;
  p.ntile = (M + 16 * MT - 1) / (16 * MT);
  const int ny = conv_cout_pad(p.Cout) / (16 * NT);
  long gx = 256 / ny;  // one workgroup per CU over all channel tiles
  if (gx < 1) gx = 1;
  if (gx * 8 > p.ntile) gx = (p.ntile + 7) / 8;
  p.xcd = (int)((tune().xcd_map >> 3) & 1) && (ny == 1 || gx % 8 == 0);
  hipLaunchKernelGGL((conv3s_kernel<NT, MT, UPT, S, NB>), dim3((unsigned)gx, (unsigned)ny, 1), dim3(512), lds, st, p);
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_conv2d(c3s): %s", hipGetErrorString(e_));
  g_last_variant = 8000 + NT * 100 + MT * 10 + (NB == 9 ? 5 : 0) + S;
  return 1;
}
template <int NT, int MT, int UPT, int NB>
static int c3s_launch1(const ConvP& p, hipStream_t st) { return p.stride == 1 ? c3s_launch2<NT, MT, UPT, 1, NB>(p, st) : c3s_launch2<NT, MT, UPT, 2, NB>(p, st); }
template <int NT, int MT, int NB>
static int c3s_launch0(const ConvP& p, hipStream_t st) {
  switch (p.srcC[0]) {
    case 64: return c3s_launch1<NT, MT, 1, NB>(p, st);
    case 128: return c3s_launch1<NT, MT, 2, NB>(p, st);
    default: return c3s_launch1<NT, MT, 4, NB>(p, st);
  }
}
template <typename T>
static int dispatch_c3s(ConvP p, int ngroup, hipStream_t st) {
  if constexpr (sizeof(T) != 2) return 0;
  else {
    if (!tune().c3s || p.k != 3 || p.nsrc != 1 || p.srcUp[0] || ngroup != 1) return 0;
    const int Cin = p.srcC[0];
    if (Cin != 64 && Cin != 128 && Cin != 256) return 0;
    const int ntp = conv_nt(p.Cout);
    if (conv_cout_pad(p.Cout) != p.Cout && conv_cout_pad(p.Cout) / 16 != ntp) return 0;
    // widest channel tile (a whole number of 16-row blocks of the packing tile) whose [16*NT][Kpad] weights fit one CU's LDS
    int nt = 0;
    const int opts[3] = {4, 2, 1};
    for (int i = 0; i < 3 && !nt; ++i)
      if (opts[i] <= ntp && ntp % opts[i] == 0 && (size_t)16 * opts[i] * p.Kpad * 2 <= 156 * 1024) nt = opts[i];
    if (!nt) return 0;
    const long npix = (long)p.B * p.H * p.W;
    const long bytes = ((npix - 1) * p.srcCs[0] + p.srcC[0]) * 2L;
    if (bytes >= (1L << 31) || (long)conv_cout_pad(p.Cout) * p.Kpad * 2L >= (1L << 31)) return 0;
    p.srcBytes[0] = (unsigned)bytes;
    p.NTpack = ntp;
    p.LSw = ws_ls(p.Kpad);
    const long M = (long)p.B * p.Ho * p.Wo;
    // Measured at batch 32 (tools/c3s_bench.sh, profiles/r03_c3s_vs_tile.txt): the stream kernel wins where the layer is big enough to
    // keep every CU streaming -- the stride-2 down-sampling convs (layer 3: 68 -> 47 us, 5: 56 -> 38, 7: 34 -> 27, 17: 19.5 -> 16) -- and
    // loses to the LDS-halo tile kernel at stride 1 (every input line goes through the vector-memory path 9 times: L2 hits, but at
    // ~30 B/clk per CU that is 13 us for the 80x80 box-tower convs) and on the smallest maps.  c3s = 2 forces it everywhere (tests).
    const long work = M * (conv_cout_pad(p.Cout) / (16 * nt));
    if (tune().c3s < 2 && (p.stride != 2 || work < tune().c3s_min_work)) return 0;
    // wave tile: 4 pixel blocks per wave on the big layers, 2 (more, smaller wave tiles) otherwise
    long cfg = tune().c3s_cfg;  // (developer knob: MT * 10 + ring depth)
    if (!cfg) cfg = (nt == 4 && M >= tune().c3s_mt4_m) ? 43 : 23;
    if (nt == 4) return cfg == 43 ? c3s_launch0<4, 4, 3>(p, st) : c3s_launch0<4, 2, 3>(p, st);
    if (nt == 2) return c3s_launch0<2, 2, 3>(p, st);
    return c3s_launch0<1, 2, 3>(p, st);
  }
}

// ---- small-M dispatch
template <typename T, int NT>
static int small_launch(ConvP p, int ngroup, hipStream_t st) {
  constexpr int BATCH = sizeof(T) == 2 ? 8 : 4;
  const long M = (long)p.B * p.Ho * p.Wo;
  p.ntile = (M + 15) / 16;
  p.ntn = conv_cout_pad(p.Cout) / (16 * NT);
  const long waves = p.ntile * p.ntn;
  hipLaunchKernelGGL((conv_small_kernel<T, NT, BATCH>), dim3((unsigned)((waves + 3) / 4), 1, ngroup), dim3(256), 0, st, p);
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_conv2d(small): %s", hipGetErrorString(e_));
  return 1;
}

static int small_pick_nt(int Cout, int es) {
  const int ntp = conv_nt(Cout);
  const int cap = es == 2 ? 5 : 2;
  const int opts[4] = {5, 4, 2, 1};
  for (int i = 0; i < 4; ++i)
    if (opts[i] <= cap && opts[i] <= ntp && ntp % opts[i] == 0) return opts[i];
  return 1;
}

// The latency-oriented kernel wins (measured) for 1x1 convs on small maps as long as the weights every wave re-reads
// from L2 stay a small total: (#16-pixel tiles) x (weight bytes) <= 48 MB.  Larger weights: weight-stationary kernel.
#define EY_SMALL_M 100000
static bool small_ok(int Cout, int Kpad, int k, long M, int es) {
  if (k != 1 || M >= tune().small_m) return false;
  return ((M + 15) / 16) * (long)conv_cout_pad(Cout) * Kpad * es <= (tune().small_wmb << 20);
}

template <typename T>
static int dispatch_small(ConvP p, int ngroup, hipStream_t st) {
  const long M = (long)p.B * p.Ho * p.Wo;
  if (!small_ok(p.Cout, p.Kpad, p.k, M, sizeof(T))) return 0;
  p.Ctot = 0; p.nsteps = 0;
  for (int s2 = 0; s2 < p.nsrc; ++s2) {
    p.Ctot += p.srcC[s2];
    p.nsteps += (p.srcC[s2] + 31) / 32;
    const int up = p.srcUp[s2];
    const long npix = (long)p.B * (p.H >> up) * (p.W >> up);
    const long bytes = ((npix - 1) * p.srcCs[s2] + p.srcC[s2]) * (long)sizeof(T);
    if (bytes >= (1L << 31) || p.srcG * (long)sizeof(T) * (ngroup - 1) >= (1L << 31)) return 0;
    p.srcBytes[s2] = (unsigned)bytes;
  }
  p.nsteps *= p.k * p.k;
  if (p.nsrc == 1) { p.srcC[1] = p.srcC[0]; p.srcCs[1] = p.srcCs[0]; p.srcUp[1] = p.srcUp[0]; p.src[1] = p.src[0]; }
  p.NTpack = conv_nt(p.Cout);
  switch (small_pick_nt(p.Cout, sizeof(T))) {
    case 5: if constexpr (sizeof(T) == 2) return small_launch<T, 5>(p, ngroup, st); else return small_launch<T, 1>(p, ngroup, st);
    case 4: if constexpr (sizeof(T) == 2) return small_launch<T, 4>(p, ngroup, st); else return small_launch<T, 2>(p, ngroup, st);
    case 2: return small_launch<T, 2>(p, ngroup, st);
    default: return small_launch<T, 1>(p, ngroup, st);
  }
}

// ---- lean pointwise dispatch

static int pw_pick_nt(int Cout, long mtiles, int es) {
  const int ntp = conv_nt(Cout), rows = conv_cout_pad(Cout) / 16;
  const int opts[5] = {8, 5, 4, 2, 1};
  int pick = 0;
  for (int i = 0; i < 5; ++i) {
    const int nt = opts[i];
    if (nt > ntp || ntp % nt || (es == 4 && nt > 4)) continue;
    pick = nt;  // candidates come widest first; keep narrowing until there are enough waves (but stay >= 2 for 16-byte stores)
    if (mtiles * (rows / nt) >= tune().pw_waves || nt <= 2) break;
  }
  return pick;
}

template <typename T, int NT>
static int pw_launch(const ConvP& p, bool two, bool geo, hipStream_t st) {
  // k-steps in flight per wave: as many as keep the wave at <= ~128 VGPRs (4 waves per SIMD)
  constexpr int BATCH = (sizeof(T) == 2 ? (NT <= 2 ? 8 : NT <= 5 ? 4 : 2) : (NT <= 2 ? 4 : 2));
  const dim3 grid((unsigned)((p.ntile + 3) / 4), (unsigned)(conv_cout_pad(p.Cout) / (16 * NT)), 1);
  if (two) {
    if (geo) hipLaunchKernelGGL((conv_pw_kernel<T, NT, BATCH, true, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_pw_kernel<T, NT, BATCH, true, false>), grid, dim3(256), 0, st, p);
  } else {
    if (geo) hipLaunchKernelGGL((conv_pw_kernel<T, NT, BATCH, false, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_pw_kernel<T, NT, BATCH, false, false>), grid, dim3(256), 0, st, p);
  }
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_conv2d(pw): %s", hipGetErrorString(e_));
  g_last_variant = 4000 + NT * 10 + (two ? 2 : 1);
  return 1;
}

// ---- register-stationary pointwise dispatch (large maps, few channels)
template <typename T, int NT, int KS, bool TWO, bool GEO>
static int pwr_launch2(const ConvP& p, hipStream_t st) {
  // persistent grid = exactly the waves that are resident at once (register-limited), tiles dealt round-robin
  static int occ = 0;
  if (!occ) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv_pwr_kernel<T, NT, KS, TWO, GEO>, 256, 0) != hipSuccess || n < 1) n = 1;
    occ = n > 4 ? 4 : n;
  }
  const unsigned ny = (unsigned)(conv_cout_pad(p.Cout) / (16 * NT));
  long gx = (long)256 * occ / ny / tune().grid_div;
  const long need = (p.ntile + 3) / 4;
  if (gx > need) gx = need;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((conv_pwr_kernel<T, NT, KS, TWO, GEO>), dim3((unsigned)gx, ny, 1), dim3(256), 0, st, p);
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_conv2d(pwr): %s", hipGetErrorString(e_));
  g_last_variant = 5000 + NT * 10 + KS;
  return 1;
}

template <typename T, int NT, int KS>
static int pwr_launch(const ConvP& p, bool two, bool geo, hipStream_t st) {
  if (two) return geo ? pwr_launch2<T, NT, KS, true, true>(p, st) : pwr_launch2<T, NT, KS, true, false>(p, st);
  return geo ? pwr_launch2<T, NT, KS, false, true>(p, st) : pwr_launch2<T, NT, KS, false, false>(p, st);
}

template <typename T, int NT>
static int pwr_launch_ks(const ConvP& p, int ks, bool two, bool geo, hipStream_t st) {
  switch (ks) {
    case 1: return pwr_launch<T, NT, 1>(p, two, geo, st);
    case 2: return pwr_launch<T, NT, 2>(p, two, geo, st);
    case 3: return pwr_launch<T, NT, 3>(p, two, geo, st);
    case 4: if constexpr (NT <= 4) return pwr_launch<T, NT, 4>(p, two, geo, st); else return 0;
  }
  return 0;
}

template <typename T>
static int dispatch_pwr(ConvP p, int ngroup, hipStream_t st) {
  const long M = (long)p.B * p.Ho * p.Wo;
  if (sizeof(T) != 2) return 0;  // f16 throughput mode only (an f32 fragment is twice the registers)
  if (p.k != 1 || p.stride != 1 || ngroup != 1 || M < tune().pwr_m || !p.vec_store || M >= (1L << 27)) return 0;
  const int ntp = conv_nt(p.Cout);
  if (conv_cout_pad(p.Cout) != 16 * ntp) return 0;  // one channel tile covers Cout (Cout <= 128)
  const int ks = (p.srcC[0] + 31) / 32 + (p.nsrc == 2 ? (p.srcC[1] + 31) / 32 : 0);
  if (ks * ntp > tune().pwr_frags || ks > 4) return 0;
  for (int s2 = 0; s2 < p.nsrc; ++s2) {
    const int up = p.srcUp[s2];
    const long npix = (long)p.B * (p.H >> up) * (p.W >> up);
    const long bytes = ((npix - 1) * p.srcCs[s2] + p.srcC[s2]) * (long)sizeof(T);
    if (bytes >= (1L << 31)) return 0;
    p.srcBytes[s2] = (unsigned)bytes;
  }
  p.ntile = (M + 15) / 16;
  p.NTpack = ntp;
  const bool two = p.nsrc == 2, geo = p.addz != nullptr || p.srcUp[0] || (two && p.srcUp[1]);
  if constexpr (sizeof(T) == 2) {
    switch (ntp) {
      case 1: return pwr_launch_ks<T, 1>(p, ks, two, geo, st);
      case 2: return pwr_launch_ks<T, 2>(p, ks, two, geo, st);
      case 4: return pwr_launch_ks<T, 4>(p, ks, two, geo, st);
      case 5: return pwr_launch_ks<T, 5>(p, ks, two, geo, st);
      case 8: return pwr_launch_ks<T, 8>(p, ks, two, geo, st);
    }
  }
  return 0;
}

template <typename T>
static int dispatch_pw(ConvP p, int ngroup, hipStream_t st) {
  const long M = (long)p.B * p.Ho * p.Wo;
  if (p.k != 1 || p.stride != 1 || ngroup != 1 || M >= tune().pw_m) return 0;
  if (((M + 15) / 16) * (long)conv_cout_pad(p.Cout) * p.Kpad * (long)sizeof(T) > (tune().pw_wmb << 20)) return 0;  // every wave re-reads its weight rows
  if ((long)conv_cout_pad(p.Cout) * p.Kpad * (long)sizeof(T) >= (1L << 31)) return 0;
  for (int s2 = 0; s2 < p.nsrc; ++s2) {
    const int up = p.srcUp[s2];
    const long npix = (long)p.B * (p.H >> up) * (p.W >> up);
    const long bytes = ((npix - 1) * p.srcCs[s2] + p.srcC[s2]) * (long)sizeof(T);
    if (bytes >= (1L << 31)) return 0;
    p.srcBytes[s2] = (unsigned)bytes;
  }
  p.ntile = (M + 15) / 16;
  p.NTpack = conv_nt(p.Cout);
  const bool two = p.nsrc == 2, geo = p.addz != nullptr || p.srcUp[0] || (two && p.srcUp[1]);
  switch (pw_pick_nt(p.Cout, p.ntile, sizeof(T))) {
    case 8: if constexpr (sizeof(T) == 2) return pw_launch<T, 8>(p, two, geo, st); else return 0;
    case 5: if constexpr (sizeof(T) == 2) return pw_launch<T, 5>(p, two, geo, st); else return 0;
    case 4: return pw_launch<T, 4>(p, two, geo, st);
    case 2: return pw_launch<T, 2>(p, two, geo, st);
    case 1: return pw_launch<T, 1>(p, two, geo, st);
  }
  return 0;
}

// ---- N-split pointwise kernel (conv_pwn_kernel): small maps, K = 128 ... 512, Cout % 128 == 0
template <int KS, int NTW>
static int pwn_launch(ConvP p, hipStream_t st) {
  int units = KS * 4;
  while ((units & 3) != 2) ++units;  // LDS pixel pitch: 2 (mod 4) 16-byte units (conflict-free fragment reads)
  p.LSw = units * 8;
  const size_t lds = (size_t)64 * p.LSw * 2;
  static bool reserved = false;
  if (lds > 64 * 1024 && !reserved) {
    if (hipFuncSetAttribute((const void*)conv_pwn_kernel<KS, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "conv(pwn): cannot reserve %zu B of LDS", lds);
    reserved = true;
  }
  const long M = (long)p.B * p.Ho * p.Wo;
  g_last_variant = 3000 + KS * 10 + NTW;
  hipLaunchKernelGGL((conv_pwn_kernel<KS, NTW>), dim3((unsigned)((M + 63) / 64), (unsigned)(p.Cout / (128 * NTW))), dim3(512), lds, st, p);
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_conv2d(pwn): %s", hipGetErrorString(e_));
  return 1;
}
template <typename T>
static int dispatch_pwn(ConvP p, int ngroup, hipStream_t st) {
  if constexpr (sizeof(T) != 2) return 0;
  else {
    const long M = (long)p.B * p.Ho * p.Wo;
    if (!tune().pwn || p.k != 1 || p.stride != 1 || ngroup != 1 || M > tune().pwn_max_m || M < 1024 || p.Cout % 128 || p.res || p.addz || p.out_scale != 1.f ||
        p.vec_store != 2 || (p.bias && !ey_aligned(p.bias, 16)))
      return 0;
    int K = 0;
    for (int s2 = 0; s2 < p.nsrc; ++s2) {
      if (p.srcC[s2] % 32) return 0;
      K += p.srcC[s2];
      const int up = p.srcUp[s2];
      const long bytes = (((long)p.B * (p.H >> up) * (p.W >> up) - 1) * p.srcCs[s2] + p.srcC[s2]) * 2;
      if (bytes >= (1L << 31)) return 0;
      p.srcBytes[s2] = (unsigned)bytes;
    }
    if (K < 128 || K > 512 || M * p.yCs * 2 >= (1L << 31)) return 0;
    p.NTpack = 8;
    p.Ctot = K;
    long ntw = tune().pwn_ntw;
    if (ntw != 1 && ntw != 2) ntw = (p.Cout % 256 == 0 && (M + 63) / 64 >= 192) ? 2 : 1;  // 256-channel slabs once the pixel tiles alone fill the chip
    if (p.Cout % 256) ntw = 1;
#define PWN(KSV) if (K == 32 * KSV) return ntw == 2 ? pwn_launch<KSV, 2>(p, st) : pwn_launch<KSV, 1>(p, st);
    PWN(4) PWN(6) PWN(8) PWN(12) PWN(16)
#undef PWN
    return 0;
  }
}

// returns 1 if launched, 0 if this shape does not fit the weight-stationary kernel, <0 on error
template <typename T>
static int dispatch_ws(ConvP p, int ngroup, hipStream_t st) {
  int nt = ws_pick_nt(p.Cout, p.Kpad, sizeof(T), (size_t)(tune().ws_lds_kb << 10));
  if (!nt) nt = ws_pick_nt(p.Cout, p.Kpad, sizeof(T), 156 * 1024);
  if (!nt) return 0;
  if (p.k == 3 && nt < tune().ws_k3_minnt && nt < conv_nt(p.Cout)) return 0;
  p.NTpack = conv_nt(p.Cout);
  p.LSw = ws_ls(p.Kpad);
  p.Ctot = 0; p.nsteps = 0;
  for (int s2 = 0; s2 < p.nsrc; ++s2) { p.Ctot += p.srcC[s2]; p.nsteps += (p.srcC[s2] + 31) / 32; }
  p.nsteps *= p.k * p.k;
  if (p.nsrc == 1) { p.srcC[1] = p.srcC[0]; p.srcCs[1] = p.srcCs[0]; p.srcUp[1] = p.srcUp[0]; }
  for (int s2 = 0; s2 < p.nsrc; ++s2) {
    const int up = p.srcUp[s2];
    const long npix = (long)p.B * (p.H >> up) * (p.W >> up);
    const long bytes = ((npix - 1) * p.srcCs[s2] + p.srcC[s2] + (ngroup - 1) * p.srcG * 0) * (long)sizeof(T);
    if (bytes >= (1L << 31) || p.srcG * (long)sizeof(T) * (ngroup - 1) >= (1L << 31)) return 0;  // beyond 32-bit buffer offsets: chunked kernel
    p.srcBytes[s2] = (unsigned)bytes;
  }
  const bool ok = p.k == 1 ? ws_launch_nt<T, 1>(p, nt, ngroup, st) : ws_launch_nt<T, 3>(p, nt, ngroup, st);
  if (!ok) return ey_set_error(EY_ELAUNCH, "conv: cannot reserve LDS for the weight tile");
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_conv2d(ws): %s", hipGetErrorString(e_));
  return 1;
}

template <typename T>
static int dispatch_conv(const ConvP& p, int ngroup, hipStream_t st) {
  bool ok;
  switch (conv_nt(p.Cout)) {
    case 1: ok = launch_conv<T, 1>(p, ngroup, st); break;
    case 2: ok = launch_conv<T, 2>(p, ngroup, st); break;
    case 4: ok = launch_conv<T, 4>(p, ngroup, st); break;
    case 5: ok = launch_conv<T, 5>(p, ngroup, st); break;
    default: ok = launch_conv<T, 8>(p, ngroup, st); break;
  }
  if (!ok) return ey_set_error(EY_ELAUNCH, "conv: cannot reserve LDS for the weight tile");
  EY_LAUNCH_CHECK("ey_conv2d");
  return EY_OK;
}

// dispatch order: lean pointwise -> small-M -> 3x3 halo tile -> weight-stationary -> K-chunked fallback
template <typename T>
static int conv2d_typed(const ConvP& p, int ngroup, hipStream_t st) {
  const int pw = dispatch_pw<T>(p, ngroup, st);
  if (pw != 0) return pw < 0 ? pw : EY_OK;
  const int pn = dispatch_pwn<T>(p, ngroup, st);
  if (pn != 0) return pn < 0 ? pn : EY_OK;
  const int pwr = dispatch_pwr<T>(p, ngroup, st);
  if (pwr != 0) return pwr < 0 ? pwr : EY_OK;
  const int sm = dispatch_small<T>(p, ngroup, st);
  if (sm != 0) return sm < 0 ? sm : EY_OK;
  const int cr = dispatch_c3r<T>(p, ngroup, st);
  if (cr != 0) return cr < 0 ? cr : EY_OK;
  const int cs = dispatch_c3s<T>(p, ngroup, st);
  if (cs != 0) return cs < 0 ? cs : EY_OK;
  const int cp = dispatch_c3p<T>(p, ngroup, st);
  if (cp != 0) return cp < 0 ? cp : EY_OK;
  const int tl = dispatch_tile<T>(p, ngroup, st);
  if (tl != 0) return tl < 0 ? tl : EY_OK;
  const int halo = dispatch_halo<T>(p, ngroup, st);
  if (halo != 0) return halo < 0 ? halo : EY_OK;
  const int ws = dispatch_ws<T>(p, ngroup, st);
  if (ws != 0) return ws < 0 ? ws : EY_OK;
  return dispatch_conv<T>(p, ngroup, st);
}

// the f16 and f32 instantiations live in two translation units (conv_f16.hip / conv_f32.hip) so that they compile in parallel
int ey_conv2d_run_f16(const ConvP& p, int ngroup, hipStream_t st);
int ey_conv2d_run_f32(const ConvP& p, int ngroup, hipStream_t st);
#if EY_CONV_PART == 32
int ey_conv2d_run_f32(const ConvP& p, int ngroup, hipStream_t st) { return conv2d_typed<float>(p, ngroup, st); }
#else
int ey_conv2d_run_f16(const ConvP& p, int ngroup, hipStream_t st) { return conv2d_typed<f16>(p, ngroup, st); }

static int conv_desc_to_p(const ey_conv_desc* d, ConvP& p, int& ngroup) {
  EY_CHECK(d, "conv: null desc");
  EY_CHECK(d->dtype == EY_F16 || d->dtype == EY_F32, "conv: bad dtype %d", d->dtype);
  const int es = d->dtype == EY_F16 ? 2 : 4;
  EY_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "conv: bad extent B=%d H=%d W=%d Cout=%d", d->B, d->H, d->W, d->Cout);
  EY_CHECK((d->k == 1 || d->k == 3) && (d->stride == 1 || d->stride == 2) && d->pad == d->k / 2,
           "conv: k=%d stride=%d pad=%d unsupported by the MFMA kernel (use ey_conv2d_direct)", d->k, d->stride, d->pad);
  EY_CHECK(d->Ho == (d->H + 2 * d->pad - d->k) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->k) / d->stride + 1,
           "conv: Ho/Wo (%d,%d) inconsistent with H/W (%d,%d)", d->Ho, d->Wo, d->H, d->W);
  EY_CHECK(d->nsrc == 1 || d->nsrc == 2, "conv: nsrc=%d", d->nsrc);
  EY_CHECK(d->w && d->y, "conv: null weight/output");
  int Cin = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    EY_CHECK(d->src[s], "conv: null src%d", s);
    EY_CHECK(d->src_C[s] > 0 && d->src_C[s] % 8 == 0, "conv: src%d channels %d not a multiple of 8 (use ey_conv2d_direct)", s, d->src_C[s]);
    EY_CHECK(d->src_cstride[s] >= d->src_C[s] && (d->src_cstride[s] * es) % 16 == 0 && ey_aligned(d->src[s], 16),
             "conv: src%d view (cstride %d) not 16-byte aligned", s, d->src_cstride[s]);
    EY_CHECK(d->src_up[s] == 0 || d->src_up[s] == 1, "conv: src_up must be 0/1");
    EY_CHECK(!d->src_up[s] || (d->H % 2 == 0 && d->W % 2 == 0), "conv: upsampled source needs even H,W");
    Cin += d->src_C[s];
  }
  EY_CHECK(d->y_cstride >= d->Cout, "conv: y_cstride %d < Cout %d", d->y_cstride, d->Cout);
  EY_CHECK(!d->res || d->res_cstride >= d->Cout, "conv: res_cstride");
  EY_CHECK(!d->addz || (d->addz_H > 0 && d->addz_W > 0 && d->addz_cstride >= d->Cout), "conv: addz extent/cstride");
  ngroup = d->ngroup > 0 ? d->ngroup : 1;
  EY_CHECK(ngroup == 1 || d->nsrc == 1, "conv: ngroup>1 needs a single source");
  p.B = d->B; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.k = d->k; p.stride = d->stride;
  p.pad = d->pad; p.act = d->act; p.nsrc = d->nsrc;
  for (int s = 0; s < 2; ++s) {
    p.src[s] = s < d->nsrc ? d->src[s] : nullptr;
    p.srcC[s] = s < d->nsrc ? d->src_C[s] : 0;
    p.srcCs[s] = s < d->nsrc ? d->src_cstride[s] : 0;
    p.srcUp[s] = s < d->nsrc ? d->src_up[s] : 0;
  }
  p.w = d->w; p.bias = d->bias; p.y = d->y; p.yCs = d->y_cstride; p.res = d->res; p.resCs = d->res_cstride;
  p.out_scale = d->out_scale; p.addz = d->addz; p.addzCs = d->addz_cstride; p.Hz = d->addz_H; p.Wz = d->addz_W;
  p.zsy = d->addz ? (float)d->addz_H / (float)d->Ho : 0.f; p.zsx = d->addz ? (float)d->addz_W / (float)d->Wo : 0.f; p.srcG = d->src_gstride; p.yG = d->y_gstride;
  p.wG = d->w_gstride; p.wGmax = d->w_gmax > 0 ? d->w_gmax : 0;
  p.Kpad = conv_kpad(Cin, d->k, es);
  p.xcd = 0;
  p.nchunks = 0;
  for (int s2 = 0; s2 < d->nsrc; ++s2) p.nchunks += (d->src_C[s2] + CONV_CH - 1) / CONV_CH;
  p.nchunks *= d->k * d->k;
  const int va = 4 * es;  // 4-element vector access alignment
  p.vec_store = d->Cout % 4 == 0 && (d->y_cstride * es) % va == 0 && ey_aligned(d->y, va) && (d->y_gstride * es) % va == 0 &&
                (!d->res || ((d->res_cstride * es) % va == 0 && ey_aligned(d->res, va))) &&
                (!d->addz || ((d->addz_cstride * es) % va == 0 && ey_aligned(d->addz, va))) && (!d->bias || ey_aligned(d->bias, 16));
  if (p.vec_store && (d->y_cstride * es) % 16 == 0 && ey_aligned(d->y, 16) && (d->y_gstride * es) % 16 == 0 &&
      (!d->res || ((d->res_cstride * es) % 16 == 0 && ey_aligned(d->res, 16))))
    p.vec_store = 2;  // 16-byte epilogue accesses allowed
  return EY_OK;
}

extern "C" int ey_conv2d(const ey_conv_desc* d, ey_stream_t stream) {
  ConvP p;
  int ngroup = 1;
  const int rc = conv_desc_to_p(d, p, ngroup);
  if (rc != EY_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  g_last_variant = 0;
  return d->dtype == EY_F16 ? ey_conv2d_run_f16(p, ngroup, st) : ey_conv2d_run_f32(p, ngroup, st);
}

// ---- two chained pointwise convs (conv_pwc_kernel): the second reads exactly what the first writes
extern "C" int ey_conv_pw_pair(const ey_conv_desc* first, const ey_conv_desc* second, ey_stream_t stream) {
  ConvP p, q;
  int g1 = 1, g2 = 1;
  int rc = conv_desc_to_p(first, p, g1);
  if (rc != EY_OK) return rc;
  rc = conv_desc_to_p(second, q, g2);
  if (rc != EY_OK) return rc;
  const long M = (long)p.B * p.Ho * p.Wo;
  const int nt = conv_nt(p.Cout);
  const bool fits = tune().pwc && first->dtype == EY_F16 && second->dtype == EY_F16 && g1 == 1 && g2 == 1 && p.k == 1 && q.k == 1 && p.stride == 1 && q.stride == 1 &&
                    p.nsrc == 1 && q.nsrc == 1 && !p.srcUp[0] && !q.srcUp[0] && (p.Cout == 64 || p.Cout == 128) && q.Cout == p.Cout && q.srcC[0] == p.Cout &&
                    p.srcC[0] % 32 == 0 && p.srcC[0] <= 128 && q.src[0] == p.y && q.srcCs[0] == p.yCs && q.B == p.B && q.H == p.Ho && q.W == p.Wo && !q.res && !q.addz &&
                    q.y != p.y && p.vec_store == 2 && q.vec_store == 2 && M <= tune().pw_m && nt * 16 == p.Cout;
  if (!fits) return ey_set_error(EY_EUNSUPPORTED, "conv_pw_pair: shapes outside the chained kernel (f16, 1x1 -> 1x1, 64 or 128 channels, small maps)");
  for (ConvP* c : {&p, &q}) {
    const long bytes = (((long)c->B * c->H * c->W - 1) * c->srcCs[0] + c->srcC[0]) * 2;
    if (bytes >= (1L << 31)) return ey_set_error(EY_EUNSUPPORTED, "conv_pw_pair: view larger than 2 GiB");
    c->srcBytes[0] = (unsigned)bytes;
    c->NTpack = nt;
    c->ntile = (M + 15) / 16;
  }
  const dim3 grid((unsigned)((p.ntile + 3) / 4));
  const bool ag = tune().pwc == 2;
  if (nt == 4) {
    if (ag) hipLaunchKernelGGL((conv_pwc_kernel<4, true>), grid, dim3(256), 0, (hipStream_t)stream, p, q);
    else hipLaunchKernelGGL((conv_pwc_kernel<4, false>), grid, dim3(256), 0, (hipStream_t)stream, p, q);
  } else {
    if (ag) hipLaunchKernelGGL((conv_pwc_kernel<8, true>), grid, dim3(256), 0, (hipStream_t)stream, p, q);
    else hipLaunchKernelGGL((conv_pwc_kernel<8, false>), grid, dim3(256), 0, (hipStream_t)stream, p, q);
  }
  EY_LAUNCH_CHECK("ey_conv_pw_pair");
  return EY_OK;
}


// ---- chained pointwise pair (see conv_pw2_kernel)
extern "C" int ey_conv_chain_kperm(int Cmid, int* perm, int perm_len) {
  const int nt1 = conv_nt(Cmid);
  EY_CHECK(Cmid == 16 * nt1 && perm, "chain_kperm: Cmid=%d must be a whole channel tile (16, 32, 64, 80, 128)", Cmid);
  const int ks2 = (4 * nt1 + 7) / 8;
  EY_CHECK(perm_len == 32 * ks2, "chain_kperm: perm_len must be %d", 32 * ks2);
  for (int s2 = 0; s2 < ks2; ++s2)
    for (int g = 0; g < 4; ++g)
      for (int j = 0; j < 8; ++j) {
        const int idx = 8 * s2 + j;
        perm[32 * s2 + 8 * g + j] = idx < 4 * nt1 ? g * 4 * nt1 + idx : -1;  // -1: zero column
      }
  return EY_OK;
}
extern "C" int ey_conv_chain_klen(int Cmid) { const int nt1 = conv_nt(Cmid); return Cmid == 16 * nt1 ? 32 * ((4 * nt1 + 7) / 8) : 0; }

template <int NT1, int KS1, int NT2>
static int pw2_launch(const ConvP& p, const ChainP& q, hipStream_t st) {
  static int occ = 0;
  if (!occ) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv_pw2_kernel<NT1, KS1, NT2>, 256, 0) != hipSuccess || n < 1) n = 1;
    occ = n > 4 ? 4 : n;
  }
  long gx = 256L * occ / tune().grid_div;
  if (gx > (p.ntile + 3) / 4) gx = (p.ntile + 3) / 4;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((conv_pw2_kernel<NT1, KS1, NT2>), dim3((unsigned)gx), dim3(256), 0, st, p, q);
  EY_LAUNCH_CHECK("ey_conv_pw_chain");
  return EY_OK;
}

extern "C" int ey_conv_pw_chain(int dtype, int B, int H, int W, int Cin, int Cmid, int Cout, const void* x, int x_cstride, const void* w1_packed, const float* b1,
                                int act1, const void* w2_packed, const float* b2, int act2, void* y, int y_cstride, ey_stream_t stream) {
  EY_CHECK(dtype == EY_F16, "conv_pw_chain: f16 only");
  EY_CHECK(x && w1_packed && w2_packed && y && B > 0 && H > 0 && W > 0, "conv_pw_chain: bad arguments");
  const int nt1 = conv_nt(Cmid), nt2 = conv_nt(Cout), ks1 = (Cin + 31) / 32;
  // two shapes of the Detect class tower (head.py:59,68-70; c3 = max(ch[0], min(nc, 100))): nc = 80 -> 80 -> 80 -> 80, and small class
  // counts (GC10-DET, nc = 10: c3 = 64) -> 64 -> 64 -> nc <= 16
  const bool wide = nt1 == 5 && Cmid == 80 && nt2 == 5 && Cout <= 80 && Cout % 4 == 0 && ks1 == 3 && Cin % 8 == 0;
  const bool narrow = nt1 == 4 && Cmid == 64 && nt2 == 1 && Cout >= 1 && Cout <= 16 && ks1 == 2 && Cin % 8 == 0;
  if (!wide && !narrow)
    return ey_set_error(EY_EUNSUPPORTED, "conv_pw_chain: built for Cin 72..96 -> 80 -> <= 80 and Cin 40..64 -> 64 -> <= 16 (got %d -> %d -> %d)", Cin, Cmid, Cout);
  EY_CHECK(x_cstride >= Cin && (x_cstride * 2) % 16 == 0 && ey_aligned(x, 16) && y_cstride >= Cout && (y_cstride * 2) % 8 == 0 && ey_aligned(y, 8), "conv_pw_chain: view alignment");
  (void)0;
  const long M = (long)B * H * W, bytes = ((M - 1) * x_cstride + Cin) * 2L;
  EY_CHECK(bytes < (1L << 31) && M < (1L << 27), "conv_pw_chain: tensor too large");
  ConvP p;
  p.B = B; p.H = H; p.W = W; p.Ho = H; p.Wo = W; p.Cout = Cmid; p.act = act1; p.nsrc = 1;
  p.src[0] = x; p.srcC[0] = Cin; p.srcCs[0] = x_cstride; p.srcBytes[0] = (unsigned)bytes;
  p.w = w1_packed; p.bias = b1; p.Kpad = conv_kpad(Cin, 1, 2); p.ntile = (M + 15) / 16;
  ChainP q;
  q.w2 = w2_packed; q.b2 = b2; q.act2 = act2; q.Cout2 = Cout; q.Kpad2 = conv_kpad(ey_conv_chain_klen(Cmid), 1, 2); q.y2 = y; q.y2Cs = y_cstride;
  return wide ? pw2_launch<5, 3, 5>(p, q, (hipStream_t)stream) : pw2_launch<4, 2, 1>(p, q, (hipStream_t)stream);
}

// Which kernel instantiation ey_conv2d launches for a shape (profiling / documentation only): kind*1000 + NT*10 + MT,
// kind 3 = conv_small_kernel<T,NT,BATCH> (last digit = BATCH), 2 = conv3_halo_kernel<T,NT,stride>,
// 1 = conv_ws_kernel<T,NT,MT,k>, 0 = conv_igemm_kernel<T,NT,MT>.
extern "C" int ey_conv_variant(int dtype, int Cout, int Cin, int k, int stride, int plain_single_source, long M, int ngroup) {
  const int es = dtype == EY_F16 ? 2 : 4, Kpad = conv_kpad(Cin, k, es);
  if (small_ok(Cout, Kpad, k, M, es)) return 3000 + small_pick_nt(Cout, es) * 10 + (es == 2 ? 8 : 4);
  if (k == 3 && plain_single_source && Cin <= 64 && Cin >= 48) {
    ConvP p;
    p.Cout = Cout; p.srcC[0] = Cin; p.Kpad = Kpad;
    const int MT = stride == 1 ? 2 : 1, HR = 7 * stride + 3, HC = (16 * MT - 1) * stride + 3;
    if ((long)HR * HC * (Cin >> 3) <= 512L * 10) {
      const int nt = es == 2 ? halo_pick_nt<f16>(p, stride) : halo_pick_nt<float>(p, stride);
      if (nt) return 2000 + nt * 10 + MT;
    }
  }
  int nt = ws_pick_nt(Cout, Kpad, es, 76 * 1024);
  if (!nt) nt = ws_pick_nt(Cout, Kpad, es, 156 * 1024);
  if (nt) return 1000 + nt * 10 + (M >= 300000 ? 2 : 1);
  nt = conv_nt(Cout);
  const int ntiles = (Cout + 16 * nt - 1) / (16 * nt);
  return nt * 10 + ((M + 127) / 128 * ntiles * (ngroup > 0 ? ngroup : 1) >= 512 ? 2 : 1);
}
extern "C" int ey_conv_pack_nt(int Cout) { return conv_nt(Cout); }
// kind*1000 + NT*10 + x of the kernel the last ey_conv2d on this thread launched; 0 when ey_conv_variant() describes it
// (kind 4 = conv_pw_kernel<T,NT,..>, x = number of sources).
extern "C" int ey_conv_last_variant(void) { return g_last_variant; }
#endif  // EY_CONV_PART
