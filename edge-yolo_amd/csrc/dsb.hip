// K2-pair — DSBottleneck.forward (reference nn/modules/block.py:1496-1503) as ONE kernel on the small maps:
//     y = x + DSConv_k2( DSConv_k1(x) ),     DSConv_k(t) = SiLU( BN( pw_1x1( dw_kxk(t) ) ) )      (conv.py:101-104)
// On the 20x20 / 40x40 maps the two register-strip launches of a pair are a few hundred short waves each and sit at 5-18 % of HBM
// (what sets their time is launch + dependent L2 round trips, not bytes).  Here a 512-thread workgroup owns a band of RB output rows
// of one image -- several workgroups per image, the first stage recomputed on the k2/2 halo rows of the band:
//   0. the (RB + 2(r1 + r2)) x (W + 2 r1) x C input tile goes to LDS once (range-checked buffer loads = the zero padding), the tile of
//      the intermediate tensor ((RB + 2 r2) x (W + 2 r2) x C) is zero-filled: positions outside the map ARE the second conv's padding;
//   1. stage 1 on every in-map pixel of the band + halo: a lane owns a strip of P pixels x 8 channels, runs the depthwise stencil out
//      of LDS (fp32 accumulation in the tap order of dsconv_strip_kernel), and the rounded f16 result in lane (strip r, channel group
//      g) IS the B fragment of the pointwise MFMA against register-resident pw weights; bias + SiLU, rounded to f16 into the LDS tile
//      -- exactly the values the two-launch form writes to HBM and reads back;
//   2. stage 2 the same way from that tile, + the residual (still in the input tile), 16-byte stores.
// x is read once (plus the halo rows of neighbouring bands), y written once; the intermediate tensor never exists.  Every operation
// is the one dsconv_strip_kernel performs, in the same order: the result is bit-identical to the two-launch form.
#include "common.h"
#include "tune.h"

struct DsbP {
  int B, H, W, act, add;
  const void* x; int xCs; unsigned xBytes;
  const void* wdw1; const void* wpw1; const float* b1;  // [k1][k1][C] f16 | ey_conv_pack_weight(C, C, 1) | [C]
  const void* wdw2; const void* wpw2; const float* b2;
  void* y; int yCs;
  int RB, Kpad;
};

#define DSB_SLACK 8  // pixels of slack behind each tile: the last strip of a row may read past W + K - 1 columns (discarded lanes)

// One DSConv stage out of an LDS tile.  Tile row t = map row src_row0 + t, tile column u = map column u - K/2 (K/2 zero columns on
// either side), pixel pitch CP.  LAST = false: result -> LDS tile `dst` (row 0 = map row dst_row0, dst_pad zero columns in front);
// LAST = true: + residual from the LDS tile `res` -> global memory.
template <int K, int KS, int P, bool LAST>
__device__ __forceinline__ void dsb_stage(const DsbP& p, int b, const f16* s_src, int srcW, int src_row0, const f16* s_w, const void* wpw, const float* bias_g,
                                          int row_lo, int row_hi, f16* s_dst, int dstW, int dst_row0, int dst_pad, const f16* s_res, int resW, int res_row0,
                                          int res_pad) {
  constexpr int C = 32 * KS, NT = 2 * KS, CP = C + 8, NX = P + K - 1;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int ch0 = g * 4 * NT;  // NT == NTpack: after the MFMA the lane owns channels ch0 .. ch0 + 4 NT of its pixel
  Vec8<f16> af[KS][NT];
  {
    const __amdgpu_buffer_rsrc_t rw = ey_rsrc(wpw, (unsigned)(16 * NT * p.Kpad * (int)sizeof(f16)));
    const unsigned wvoff = (unsigned)((r * p.Kpad + 8 * g) * (int)sizeof(f16));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) BufLoad8<f16>::load(af[ks][nt], rw, wvoff, (nt * 16 * p.Kpad + ks * 32) * (int)sizeof(f16));
  }
  float bias[4 * NT];
#pragma unroll
  for (int i = 0; i < 4 * NT; ++i) bias[i] = bias_g ? bias_g[ch0 + i] : 0.f;
  const int WS = (p.W + P - 1) / P;
  const int nstrip = (row_hi - row_lo) * WS;
  for (int unit = wave; unit * 16 < nstrip; unit += nwave) {
    const int si = unit * 16 + r;
    const bool sv = si < nstrip;
    const int sic = sv ? si : nstrip - 1;  // lanes beyond the band compute the last strip again and store nothing
    const int ry = sic / WS, x0 = (sic - ry * WS) * P, oy = row_lo + ry;
    Vec8<f16> bf[KS][P];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cb = ks * 32 + 8 * g;
      float acc[P][8];
#pragma unroll
      for (int q = 0; q < P; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[q][i] = 0.f;
      const f16* base = s_src + ((oy - K / 2 - src_row0) * srcW + x0) * CP + cb;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        Vec8<f16> xr[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) xr[j].load(base + (ky * srcW + j) * CP);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          Vec8<f16> w;
          w.load(s_w + (ky * K + kx) * C + cb);
#pragma unroll
          for (int q = 0; q < P; ++q) ey_fma8_mix(xr[q + kx], w, acc[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < P; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) bf[ks][q].set(i, acc[q][i]);
    }
#pragma unroll
    for (int q = 0; q < P; ++q) {
      f32x4 pacc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) pacc[nt] = (f32x4)0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks][nt].v, bf[ks][q].v, pacc[nt], 0, 0, 0);
      const int ox = x0 + q;
      if (!sv || ox >= p.W) continue;
      float v[4 * NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * nt + j] = pacc[nt][j] + bias[4 * nt + j];
      ey_act_n(v, p.act);
      if constexpr (!LAST) {
        f16* d = s_dst + ((oy - dst_row0) * dstW + ox + dst_pad) * CP + ch0;
#pragma unroll
        for (int h = 0; h < NT / 2; ++h) {
          Vec8<f16> o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o.set(j, v[8 * h + j]);
          o.store(d + 8 * h);
        }
      } else {
        const long m = ((long)b * p.H + oy) * p.W + ox;
        f16* yp = (f16*)p.y + m * p.yCs + ch0;
        const f16* rp = s_res + ((oy - res_row0) * resW + ox + res_pad) * CP + ch0;
#pragma unroll
        for (int h = 0; h < NT / 2; ++h) {
          if (p.add) {
            Vec8<f16> rr;
            rr.load(rp + 8 * h);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[8 * h + j] += rr.get(j);
          }
          Vec8<f16> o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o.set(j, v[8 * h + j]);
          o.store(yp + 8 * h);
        }
      }
    }
  }
}

template <int K1, int K2, int KS, int P1, int P2>
__global__ __launch_bounds__(512) void dsb_pair_kernel(DsbP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int C = 32 * KS, CP = C + 8, CV = C / 8, R1 = K1 / 2, R2 = K2 / 2, PPI = 512 / CV;
  const int RI = p.RB + 2 * (R1 + R2), WI = p.W + 2 * R1, RM = p.RB + 2 * R2, WM = p.W + 2 * R2;
  f16* s_in = reinterpret_cast<f16*>(smem);
  f16* s_mid = s_in + (RI * WI + DSB_SLACK) * CP;
  f16* s_w1 = s_mid + (RM * WM + DSB_SLACK) * CP;
  f16* s_w2 = s_w1 + K1 * K1 * C;
  const int tid = threadIdx.x, b = blockIdx.y, y0 = blockIdx.x * p.RB;
  {  // ---- 0. input tile (requests first), then the zero fill and the depthwise weights while they fly
    const __amdgpu_buffer_rsrc_t rx = ey_rsrc(p.x, p.xBytes);
    const int cv = tid % CV, NP = RI * WI, iy0 = y0 - R1 - R2;
    constexpr int U = 4;
    int base = tid / CV;
    bool first = true;
    do {  // (at least one round per thread: the LDS-only work below hangs on it)
      Vec8<f16> t[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int px = base + j * PPI, trow = px / WI, u = px - trow * WI, iy = iy0 + trow, ix = u - R1;
        const bool ok = px < NP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        BufLoad8<f16>::load(t[j], rx, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * p.xCs + cv * 8) * (int)sizeof(f16)) : EY_OOB);
      }
      if (first) {  // first round of requests is out: LDS-only work under their latency
        first = false;
        Vec8<f16> z;
        z.zero();
        for (int v = tid; v < (RM * WM + DSB_SLACK) * CP / 8; v += 512) z.store(s_mid + v * 8);
        for (int v = tid; v < K1 * K1 * CV; v += 512) {
          Vec8<f16> w;
          w.load((const f16*)p.wdw1 + v * 8);
          w.store(s_w1 + v * 8);
        }
        for (int v = tid; v < K2 * K2 * CV; v += 512) {
          Vec8<f16> w;
          w.load((const f16*)p.wdw2 + v * 8);
          w.store(s_w2 + v * 8);
        }
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int px = base + j * PPI;
        if (px < NP) t[j].store(s_in + px * CP + cv * 8);
      }
      base += U * PPI;
    } while (base < NP);
  }
  __syncthreads();
  {  // ---- 1. first DSConv on the band + the second conv's halo rows (in-map pixels only: the rest of the tile stays zero)
    const int lo = y0 - R2 < 0 ? 0 : y0 - R2, hi = y0 + p.RB + R2 > p.H ? p.H : y0 + p.RB + R2;
    dsb_stage<K1, KS, P1, false>(p, b, s_in, WI, y0 - R1 - R2, s_w1, p.wpw1, p.b1, lo, hi, s_mid, WM, y0 - R2, R2, nullptr, 0, 0, 0);
  }
  __syncthreads();
  {  // ---- 2. second DSConv + residual
    const int hi = y0 + p.RB > p.H ? p.H : y0 + p.RB;
    dsb_stage<K2, KS, P2, true>(p, b, s_mid, WM, y0 - R2, s_w2, p.wpw2, p.b2, y0, hi, nullptr, 0, 0, 0, s_in, WI, y0 - R1 - R2, R1);
  }
}

static size_t dsb_lds_bytes(int W, int C, int k1, int k2, int RB) {
  const int CP = C + 8, r1 = k1 / 2, r2 = k2 / 2;
  return ((size_t)((RB + 2 * (r1 + r2)) * (W + 2 * r1) + DSB_SLACK) + (size_t)((RB + 2 * r2) * (W + 2 * r2) + DSB_SLACK)) * CP * 2 + (size_t)(k1 * k1 + k2 * k2) * C * 2;
}

// Rows per band: the cheapest (rounds of workgroups over the CUs) x (stage-1 rows incl. halo x k1^2 + band rows x k2^2 + fixed cost)
static int dsb_band_rows(int B, int H, int W, int C, int k1, int k2) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu < 1) ncu = 256;
  }
  long force = tune().dsb_rb;
  if (force > H) force = H;
  if (force > 0) return dsb_lds_bytes(W, C, k1, k2, (int)force) <= 160 * 1024 ? (int)force : 0;
  int best = 0;
  double best_cost = 0;
  for (int rb = 1; rb <= H; ++rb) {
    if (dsb_lds_bytes(W, C, k1, k2, rb) > 160 * 1024) break;
    const long wgs = (long)B * ((H + rb - 1) / rb);
    const double rounds = (double)((wgs + ncu - 1) / ncu);
    const double cost = rounds * ((rb + 2 * (k2 / 2)) * k1 * k1 + rb * k2 * k2 + (double)tune().dsb_fixed);
    if (!best || cost < best_cost) best = rb, best_cost = cost;
  }
  return best;
}

template <int K1, int K2, int KS, int P1, int P2>
static int dsb_launch(DsbP p, int bands, size_t lds, hipStream_t st) {
  static size_t reserved = 0;
  if (lds > 64 * 1024 && lds > reserved) {
    if (hipFuncSetAttribute((const void*)dsb_pair_kernel<K1, K2, KS, P1, P2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "dsb_pair: cannot reserve %zu B of LDS", lds);
    reserved = lds;
  }
  hipLaunchKernelGGL((dsb_pair_kernel<K1, K2, KS, P1, P2>), dim3((unsigned)bands, (unsigned)p.B), dim3(512), lds, st, p);
  EY_LAUNCH_CHECK("ey_dsb_pair");
  return EY_OK;
}

extern "C" int ey_dsb_pair(int dtype, int B, int H, int W, int C, int k1, int k2, int act, const void* x, int x_cstride, const void* w_dw1_kkc,
                           const void* w_pw1_packed, const float* bias1, const void* w_dw2_kkc, const void* w_pw2_packed, const float* bias2, int add_residual,
                           void* y, int y_cstride, ey_stream_t stream) {
  EY_CHECK(x && y && w_dw1_kkc && w_pw1_packed && w_dw2_kkc && w_pw2_packed, "dsb_pair: null pointer");
  EY_CHECK(B > 0 && H > 0 && W > 0, "dsb_pair: bad extent");
  EY_CHECK(x_cstride >= C && y_cstride >= C, "dsb_pair: cstride");
  const int es = 2;
  const long xbytes = (((long)B * H * W - 1) * x_cstride + C) * es, ybytes = (((long)B * H * W - 1) * y_cstride + C) * es;
  const bool fits = tune().dsb_pair && dtype == EY_F16 && (C == 32 || C == 64) && k1 == 3 && (k2 == 5 || k2 == 7) && B <= 65535 && xbytes < (1L << 31) &&
                    ybytes < (1L << 31) && (x_cstride * es) % 16 == 0 && ey_aligned(x, 16) && (y_cstride * es) % 16 == 0 && ey_aligned(y, 16) &&
                    ey_aligned(w_dw1_kkc, 16) && ey_aligned(w_dw2_kkc, 16) && ey_aligned(w_pw1_packed, 16) && ey_aligned(w_pw2_packed, 16) &&
                    (long)B * H * W <= tune().dsb_max_px;
  if (!fits) return ey_set_error(EY_EUNSUPPORTED, "dsb_pair: shape outside the fused kernel (f16, C 32/64, k 3 -> 5/7, small maps)");
  const int RB = dsb_band_rows(B, H, W, C, k1, k2);
  if (RB < 1) return ey_set_error(EY_EUNSUPPORTED, "dsb_pair: a one-row band of a %d-wide map does not fit LDS", W);
  DsbP p;
  p.B = B; p.H = H; p.W = W; p.act = act; p.add = add_residual;
  p.x = x; p.xCs = x_cstride; p.xBytes = (unsigned)xbytes;
  p.wdw1 = w_dw1_kkc; p.wpw1 = w_pw1_packed; p.b1 = bias1; p.wdw2 = w_dw2_kkc; p.wpw2 = w_pw2_packed; p.b2 = bias2;
  p.y = y; p.yCs = y_cstride;
  p.RB = RB; p.Kpad = ey_conv_kpad(C, es);
  const int bands = (H + RB - 1) / RB;
  const size_t lds = dsb_lds_bytes(W, C, k1, k2, RB);
  hipStream_t st = (hipStream_t)stream;
  // strip length of the second stage: short strips give the few rows of a band enough wave-units (the first stage keeps 2)
  // (measured, tools/dsb_bench.py: C64 20x20 11.5 -> 10.5 us with single pixels, the 40x40 maps 3-10 % faster with pairs; 4 never wins)
  int P2 = (int)tune().dsb_p2;
  if (P2 != 1 && P2 != 2 && P2 != 4) P2 = RB * ((W + 1) / 2) < 64 ? 1 : 2;
#define DSB(KV, KSV)                                                          \
  if (k2 == KV && C == 32 * KSV) {                                            \
    if (P2 == 1) return dsb_launch<3, KV, KSV, 2, 1>(p, bands, lds, st);      \
    if (P2 == 4) return dsb_launch<3, KV, KSV, 2, 4>(p, bands, lds, st);      \
    return dsb_launch<3, KV, KSV, 2, 2>(p, bands, lds, st);                   \
  }
  DSB(5, 1) DSB(7, 1) DSB(5, 2) DSB(7, 2)
#undef DSB
  return ey_set_error(EY_EINVAL, "dsb_pair: k2=%d C=%d", k2, C);
}
