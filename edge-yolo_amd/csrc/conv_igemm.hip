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
// Both operands are fragment-shaped global loads (weights are tiny and L1/L2 resident; every conv on this path
// is HBM-bound on activations, which are read once because one block covers all of Cout up to 128).
#include "common.h"

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
  int Kpad;      // packed row length (elements)
  int vec_store; // 1: y/res/addz views are aligned for 4-element vector access
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

template <typename T, int NT, int MT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long M = (long)p.B * p.Ho * p.Wo;
  const long m_wave0 = ((long)blockIdx.x * 4 + wave) * (16 * MT);
  const int n_base = blockIdx.y * (16 * NT);
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

  const T* wrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wrow[nt] = (const T*)p.w + (long)(n_base + nt * 16 + r) * p.Kpad + 8 * g;

  int kofs = 0;
  for (int ky = 0; ky < p.k; ++ky) {
    for (int kx = 0; kx < p.k; ++kx) {
      int iy[MT], ix[MT];
      bool inb[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        iy[mt] = poy[mt] * p.stride - p.pad + ky;
        ix[mt] = pox[mt] * p.stride - p.pad + kx;
        inb[mt] = pv[mt] && iy[mt] >= 0 && iy[mt] < p.H && ix[mt] >= 0 && ix[mt] < p.W;
      }
      for (int s = 0; s < p.nsrc; ++s) {
        const int Cs = p.srcC[s], cs = p.srcCs[s], up = p.srcUp[s];
        const int Hs = p.H >> up, Ws = p.W >> up;
        const T* xb[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          long pix = ((long)pb[mt] * Hs + (inb[mt] ? (iy[mt] >> up) : 0)) * Ws + (inb[mt] ? (ix[mt] >> up) : 0);
          xb[mt] = (const T*)p.src[s] + (long)grp * p.srcG + pix * cs + 8 * g;
        }
        for (int c0 = 0; c0 < Cs; c0 += 32) {
          const bool cvalid = (c0 + 8 * g) < Cs;
          Vec8<T> bf[MT], af[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (inb[mt] && cvalid) bf[mt].load(xb[mt] + c0);
            else bf[mt].zero();
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) af[nt].load(wrow[nt] + kofs + c0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma16(af[nt], bf[mt], acc[mt][nt]);
        }
        kofs += Cs;
      }
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
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i)
        if (ch0 + i < p.Cout) v[i] += p.bias[ch0 + i];
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
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) v[i] = ey_act(v[i], p.act) * p.out_scale;
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
static int conv_kpad(int Cin, int k) { return k * k * Cin + 32; }  // +32: masked tail lanes may read past a row

extern "C" size_t ey_conv_packed_bytes(int dtype, int Cout, int Cin, int k) {
  return (size_t)conv_cout_pad(Cout) * conv_kpad(Cin, k) * (dtype == EY_F16 ? 2 : 4);
}

extern "C" int ey_conv_pack_weight(int dtype, int Cout, int Cin, int k, const float* w, void* out, size_t out_bytes) {
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "pack: bad dtype %d", dtype);
  EY_CHECK(Cout > 0 && Cin > 0 && (k == 1 || k == 3), "pack: Cout=%d Cin=%d k=%d", Cout, Cin, k);
  EY_CHECK(out_bytes >= ey_conv_packed_bytes(dtype, Cout, Cin, k), "pack: output buffer too small");
  const int NT = conv_nt(Cout), BN = 16 * NT, Kp = conv_kpad(Cin, k), rows = conv_cout_pad(Cout);
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

// which instantiation ey_conv2d picks: returns NT*16 + MT (profiling / documentation only)
extern "C" int ey_conv_tile(int Cout, long M, int ngroup) {
  const int NT = conv_nt(Cout), ntiles = (Cout + 16 * NT - 1) / (16 * NT);
  const long blocks2 = (M + 127) / 128 * ntiles * (ngroup > 0 ? ngroup : 1);
  return NT * 16 + (blocks2 >= 512 ? 2 : 1);
}

template <typename T, int NT>
static void launch_conv(const ConvP& p, int ngroup, hipStream_t st) {
  const long M = (long)p.B * p.Ho * p.Wo;
  const int ntiles = (p.Cout + 16 * NT - 1) / (16 * NT);
  const long blocks2 = (M + 127) / 128 * ntiles * ngroup;
  if (blocks2 >= 512) {
    dim3 grid((unsigned)((M + 127) / 128), ntiles, ngroup);
    hipLaunchKernelGGL((conv_igemm_kernel<T, NT, 2>), grid, dim3(256), 0, st, p);
  } else {
    dim3 grid((unsigned)((M + 63) / 64), ntiles, ngroup);
    hipLaunchKernelGGL((conv_igemm_kernel<T, NT, 1>), grid, dim3(256), 0, st, p);
  }
}

template <typename T>
static int dispatch_conv(const ConvP& p, int ngroup, hipStream_t st) {
  switch (conv_nt(p.Cout)) {
    case 1: launch_conv<T, 1>(p, ngroup, st); break;
    case 2: launch_conv<T, 2>(p, ngroup, st); break;
    case 4: launch_conv<T, 4>(p, ngroup, st); break;
    case 5: launch_conv<T, 5>(p, ngroup, st); break;
    default: launch_conv<T, 8>(p, ngroup, st); break;
  }
  EY_LAUNCH_CHECK("ey_conv2d");
  return EY_OK;
}

extern "C" int ey_conv2d(const ey_conv_desc* d, ey_stream_t stream) {
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
  const int ngroup = d->ngroup > 0 ? d->ngroup : 1;
  EY_CHECK(ngroup == 1 || d->nsrc == 1, "conv: ngroup>1 needs a single source");
  ConvP p;
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
  p.Kpad = conv_kpad(Cin, d->k);
  const int va = 4 * es;  // 4-element vector access alignment
  p.vec_store = d->Cout % 4 == 0 && (d->y_cstride * es) % va == 0 && ey_aligned(d->y, va) && (d->y_gstride * es) % va == 0 &&
                (!d->res || ((d->res_cstride * es) % va == 0 && ey_aligned(d->res, va)));
  hipStream_t st = (hipStream_t)stream;
  return d->dtype == EY_F16 ? dispatch_conv<f16>(p, ngroup, st) : dispatch_conv<float>(p, ngroup, st);
}
