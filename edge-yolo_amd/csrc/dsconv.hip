// K2 — fused depthwise-separable convolution (DSConv.forward, reference nn/modules/conv.py:101-104):
//     y = res + SiLU( BN( pw_1x1( dw_kxk(x) ) ) )        k in {3,5,7}, stride 1, pad k/2, BN folded into pw.
// One persistent 256-thread workgroup walks 8x16-pixel output tiles:
//   1. the (8+k-1) x (16+k-1) x Cin input halo tile is loaded ONCE into LDS (range-checked buffer loads give the
//      zero padding for free);
//   2. the depthwise stencil runs on the VALU out of LDS (fp32 accumulate, 2-pixel register strips so that every LDS
//      vector feeds two outputs) and its result is written, rounded to the storage type exactly as the reference's
//      intermediate tensor is, into an LDS pixel tile;
//   3. the pointwise 1x1 is an MFMA GEMM over that tile against pw weights staged in LDS once per workgroup, with the
//      same packed-weight layout and wide-store epilogue as the dense conv (bias, SiLU, residual).
// The intermediate (B,C,H,W) tensor of the unfused form never touches HBM: x is read once, y written once.
#include "common.h"
#include "tune.h"

static thread_local int g_ds_variant = 0;  // kernel the last ey_dsconv / ey_dsconv_tz launched: 1 LDS tile, 2 register strip, 3 Toeplitz MFMA
struct DsP {
  int B, H, W, Cin, Cout, act;
  const void* x; int xCs; unsigned xBytes;
  const void* wdw;   // [k][k][Cin] storage type
  const float* dwbias; int dwact;  // optional bias + activation between the depthwise and the pointwise conv (DWConv -> Conv towers)
  const void* wpw;   // packed by ey_conv_pack_weight(Cout, Cin, 1)
  const float* bias; // [Cout] (folded BN) or null
  void* y; int yCs;
  const void* res; int resCs;
  int Kpad, NTpack, tilesX, tilesY;
  long ntile;
  int vec_store;
  int xcd;  // Toeplitz kernel: XCD-contiguous tile order
};

__device__ __forceinline__ f32x4 ds_mma16(const Vec8<f16>& a, const Vec8<f16>& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 ds_mma16(const Vec8<float>& a, const Vec8<float>& b, f32x4 c) {
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[j], b.hi[j], c, 0, 0, 0);
  return c;
}

#define DS_TH 8
#define DS_TW 16

template <typename T, int K, int NT>
__global__ __launch_bounds__(256) void dsconv_kernel(DsP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HH = DS_TH + K - 1, HW = DS_TW + K - 1, BN = 16 * NT;
  const int C = p.Cin, CV = C >> 3;
  const int LSd = C + 8;       // dw-output tile row stride (elements)
  const int LSw = p.Kpad + (((p.Kpad >> 3) & 1) ? 0 : 8);
  float* s_in = reinterpret_cast<float*>(smem);  // [HH][HW][C]  fp32 (converted once on the way in)
  float* s_wdw = s_in + HH * HW * C;             // [K*K][C]     fp32
  T* s_dw = reinterpret_cast<T*>(s_wdw + K * K * C);  // [128][LSd]
  T* s_wpw = s_dw + DS_TH * DS_TW * LSd;         // [BN][LSw]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int n_base = blockIdx.y * BN;

  {  // weights: once per workgroup
    const T* wg = (const T*)p.wpw + (long)n_base * p.Kpad;
    const int kv = p.Kpad >> 3;
    for (int v = tid; v < BN * kv; v += 256) {
      const int row = v / kv, c8 = (v - row * kv) << 3;
      Vec8<T> w;
      w.load(wg + (long)row * p.Kpad + c8);
      w.store(s_wpw + row * LSw + c8);
    }
    for (int v = tid; v < K * K * C; v += 256) s_wdw[v] = to_f(((const T*)p.wdw)[v]);
  }
  const __amdgpu_buffer_rsrc_t rx = ey_rsrc(p.x, p.xBytes);
  const int BNp = 16 * p.NTpack;
  const int ch0 = (n_base / BNp) * BNp + g * 4 * p.NTpack + 4 * ((n_base % BNp) >> 4);
  const int tiles_img = p.tilesX * p.tilesY;

  for (long tile = blockIdx.x; tile < p.ntile; tile += gridDim.x) {
    const int b = (int)(tile / tiles_img), trem = (int)(tile - (long)b * tiles_img);
    const int ty0 = (trem / p.tilesX) * DS_TH, tx0 = (trem % p.tilesX) * DS_TW;
    __syncthreads();  // previous tile's MFMA reads of s_dw / stencil reads of s_in are done (and weights are staged)
    // ---- 1. halo tile -> LDS
    for (int v = tid; v < HH * HW * CV; v += 256) {
      const int cv = v % CV, px = v / CV, hx = px % HW, hy = px / HW;
      const int iy = ty0 + hy - K / 2, ix = tx0 + hx - K / 2;
      const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      Vec8<T> t;
      BufLoad8<T>::load(t, rx, ok ? (unsigned)(((((long)b * p.H + iy) * p.W + ix) * p.xCs + cv * 8) * (long)sizeof(T)) : EY_OOB);
      float* d = s_in + px * C + cv * 8;
      const f32x4 lo = {t.get(0), t.get(1), t.get(2), t.get(3)}, hi = {t.get(4), t.get(5), t.get(6), t.get(7)};
      *reinterpret_cast<f32x4*>(d) = lo;
      *reinterpret_cast<f32x4*>(d + 4) = hi;
    }
    __syncthreads();
    // ---- 2. depthwise stencil on fp32 LDS data: item = (row, 2-pixel strip, 4-channel group)
    {
      const int C4 = C >> 2;
      for (int it = tid; it < DS_TH * (DS_TW / 2) * C4; it += 256) {
        const int c4 = it % C4, sp = it / C4, xs = (sp % (DS_TW / 2)) * 2, y = sp / (DS_TW / 2);
        f32x4 a0 = (f32x4)0.f, a1 = (f32x4)0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const float* row = s_in + ((y + ky) * HW + xs) * C + c4 * 4;
          const float* wr = s_wdw + ky * K * C + c4 * 4;
          f32x4 prev = *reinterpret_cast<const f32x4*>(row);
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            const f32x4 nxt = *reinterpret_cast<const f32x4*>(row + (kx + 1) * C);
            const f32x4 w = *reinterpret_cast<const f32x4*>(wr + kx * C);
            a0 += prev * w;  // output pixel xs   reads column xs+kx
            a1 += nxt * w;   // output pixel xs+1 reads column xs+kx+1
            prev = nxt;
          }
        }
        T* d0 = s_dw + (y * DS_TW + xs) * LSd + c4 * 4;
        T* d1 = d0 + LSd;
        if (p.dwbias) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(p.dwbias + c4 * 4);
          a0 += bb;
          a1 += bb;
        }
        float a01[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        ey_act_n(a01, p.dwact);
#pragma unroll
        for (int i = 0; i < 4; ++i) { d0[i] = from_f<T>(a01[i]); d1[i] = from_f<T>(a01[4 + i]); }
      }
    }
    __syncthreads();
    // ---- 3. pointwise GEMM: wave w owns tile rows 2w, 2w+1 (two 16-pixel m-blocks)
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4)0.f;
    for (int c0 = 0; c0 < C; c0 += 32) {
      const bool cok = (c0 + 8 * g) < C;
      Vec8<T> bf[2], af[NT];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (cok) bf[mt].load(s_dw + ((2 * wave + mt) * DS_TW + r) * LSd + c0 + 8 * g);
        else bf[mt].zero();
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) af[nt].load(s_wpw + (nt * 16 + r) * LSw + c0 + 8 * g);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = ds_mma16(af[nt], bf[mt], acc[mt][nt]);
    }
    // ---- epilogue: lane owns pixel (row 2w+mt, column r), channels ch0 .. ch0+4NT
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int oy = ty0 + 2 * wave + mt, ox = tx0 + r;
      if (oy >= p.H || ox >= p.W) continue;
      const long m = ((long)b * p.H + oy) * p.W + ox;
      float v[4 * NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * nt + j] = acc[mt][nt][j];
      const bool full = ch0 + 4 * NT <= p.Cout;
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i) {
        const float bb = (p.bias && ch0 + i < p.Cout) ? p.bias[ch0 + i] : 0.f;
        v[i] += bb;
      }
      ey_act_n(v, p.act);
      T* yp = (T*)p.y + m * p.yCs + ch0;
      const T* rp = p.res ? (const T*)p.res + m * p.resCs + ch0 : nullptr;
      if (p.vec_store > 1 && full && sizeof(T) == 2 && NT % 2 == 0) {  // 16-byte accesses (see conv_igemm.hip epilogue)
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
      } else if (p.vec_store && full) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          float o[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
          if (rp) {
            if constexpr (sizeof(T) == 2) {
              const f16x4 rr = *reinterpret_cast<const f16x4*>(rp + 4 * q);
#pragma unroll
              for (int j = 0; j < 4; ++j) o[j] += (float)rr[j];
            } else {
              const f32x4 rr = *reinterpret_cast<const f32x4*>(rp + 4 * q);
#pragma unroll
              for (int j = 0; j < 4; ++j) o[j] += rr[j];
            }
          }
          if constexpr (sizeof(T) == 2) {
            const f16x4 ov = {(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
            *reinterpret_cast<f16x4*>(yp + 4 * q) = ov;
          } else {
            const f32x4 ov = {o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(yp + 4 * q) = ov;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4 * NT; ++i)
          if (ch0 + i < p.Cout) yp[i] = from_f<T>(v[i] + (rp ? to_f(rp[i]) : 0.f));
      }
    }
  }
}

// ================================================================================================================
// Register-strip variant (f16 throughput mode): NO activation LDS, no barriers after the weight staging.
// A lane owns a strip of P=4 consecutive output pixels x 8 channels; lane = (strip r of 16, channel group g of 4), i.e. a
// wave covers 16 strips x 32 channels per channel step.  The depthwise stencil runs row by row out of registers: K+P-1
// range-checked 16-byte buffer loads per input row (zero padding for free), each loaded vector feeding up to K*... outputs,
// fp32 accumulation, depthwise weights broadcast from a few KB of LDS.  The rounded f16 result of (pixel, 8 channels) in
// lane (r, g) IS the MFMA B fragment of the pointwise GEMM (k = 32 channels per step, column = pixel), so the 1x1 conv
// follows directly from registers against register-resident packed pw weights; epilogue = bias, SiLU, residual, wide store.
template <int K, int NT, int KS, int P>
__global__ __launch_bounds__(256) void dsconv_strip_kernel(DsP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef f16 T;
  constexpr int NX = P + K - 1;
  T* s_wdw = reinterpret_cast<T*>(smem);  // [K*K][C]
  const int C = p.Cin;
  for (int v = threadIdx.x; v < K * K * (C >> 3); v += 256) {
    Vec8<T> t;
    t.load((const T*)p.wdw + v * 8);
    t.store(s_wdw + v * 8);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int WS = (p.W + P - 1) / P;
  const int nstrip = p.B * p.H * WS;
  const int strip = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + r;  // (WS below uses the template P)
  const bool sv = strip < nstrip;
  int b = 0, oy = 0, x0 = 0;
  if (sv) {
    const int t = strip / WS;
    x0 = (strip - t * WS) * P;
    b = t / p.H;
    oy = t - b * p.H;
  }
  if ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 >= nstrip) return;  // whole wave beyond the image
  const __amdgpu_buffer_rsrc_t rx = ey_rsrc(p.x, p.xBytes);
  // pointwise weights -> registers
  Vec8<T> af[KS][NT];
  {
    const __amdgpu_buffer_rsrc_t rw = ey_rsrc(p.wpw, (unsigned)(16 * NT * p.Kpad * (int)sizeof(T)));
    const unsigned wvoff = (unsigned)((r * p.Kpad + 8 * g) * (int)sizeof(T));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) BufLoad8<T>::load(af[ks][nt], rw, wvoff, (nt * 16 * p.Kpad + ks * 32) * (int)sizeof(T));
  }
  Vec8<T> bf[KS][P];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int cb = ks * 32 + 8 * g;       // this lane's 8 channels in this step
    const bool cok = cb < C;              // lanes beyond C carry zeros into the MFMA
    float acc[P][8];
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[q][i] = (p.dwbias && cok) ? p.dwbias[cb + i] : 0.f;
    // one input row at a time (rolled loop: only two rows of vectors are ever live), the next row's loads in flight
    auto load_row = [&](int ky, Vec8<T> (&xr)[NX]) {
      const int iy = oy - K / 2 + ky;
      const bool yok = sv && cok && iy >= 0 && iy < p.H;
      const int rowoff = ((b * p.H + iy) * p.W + x0 - K / 2) * p.xCs + cb;
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        const int ix = x0 - K / 2 + j;
        BufLoad8<T>::load(xr[j], rx, (yok && ix >= 0 && ix < p.W) ? (unsigned)((rowoff + j * p.xCs) * (int)sizeof(T)) : EY_OOB);
      }
    };
    auto mac_row = [&](int ky, const Vec8<T> (&xr)[NX]) {
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        Vec8<T> w;
        if (cok) w.load(s_wdw + (ky * K + kx) * C + cb);
        else w.zero();
#pragma unroll
        for (int q = 0; q < P; ++q) ey_fma8_mix(xr[q + kx], w, acc[q]);
      }
    };
    if constexpr (K * NX <= 16) {
      // small stencils (3x3 with short strips): ALL K*NX loads in flight at once -- one memory round trip per channel step instead of K
      Vec8<T> xall[K][NX];
#pragma unroll
      for (int ky = 0; ky < K; ++ky) load_row(ky, xall[ky]);
#pragma unroll
      for (int ky = 0; ky < K; ++ky) mac_row(ky, xall[ky]);
    } else {
      Vec8<T> xin[NX], xnx[NX];
      load_row(0, xin);
#pragma unroll 1
      for (int ky = 0; ky < K; ++ky) {
        if (ky + 1 < K) load_row(ky + 1, xnx);
        mac_row(ky, xin);
#pragma unroll
        for (int j = 0; j < NX; ++j) xin[j] = xnx[j];
      }
    }
#pragma unroll
    for (int q = 0; q < P; ++q) {
      ey_act_n(acc[q], p.dwact);
#pragma unroll
      for (int i = 0; i < 8; ++i) bf[ks][q].set(i, cok ? acc[q][i] : 0.f);
    }
  }
  // ---- pointwise GEMM + epilogue, one pixel block (16 strips' q-th pixel) at a time
  const int ch0 = g * 4 * NT;  // NT == NTpack: lane owns channels ch0 .. ch0 + 4NT of its pixel
  float bias[4 * NT];
#pragma unroll
  for (int i = 0; i < 4 * NT; ++i) bias[i] = (p.bias && ch0 + i < p.Cout) ? p.bias[ch0 + i] : 0.f;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    f32x4 pacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) pacc[nt] = (f32x4)0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) pacc[nt] = ds_mma16(af[ks][nt], bf[ks][q], pacc[nt]);
    const int ox = x0 + q;
    if (!sv || ox >= p.W) continue;
    const long m = ((long)b * p.H + oy) * p.W + ox;
    float v[4 * NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * nt + j] = pacc[nt][j] + bias[4 * nt + j];
    ey_act_n(v, p.act);
    T* yp = (T*)p.y + m * p.yCs + ch0;
    const T* rp = p.res ? (const T*)p.res + m * p.resCs + ch0 : nullptr;
    if (NT % 2 == 0 && p.vec_store > 1 && ch0 + 4 * NT <= p.Cout) {
#pragma unroll
      for (int h = 0; h < NT / 2; ++h) {
        if (rp) {
          Vec8<T> rr;
          rr.load(rp + 8 * h);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[8 * h + j] += rr.get(j);
        }
        Vec8<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.set(j, v[8 * h + j]);
        o.store(yp + 8 * h);
      }
    } else {
#pragma unroll
      for (int h = 0; h < NT; ++h) {
        if (ch0 + 4 * h + 4 <= p.Cout) {
          float o[4] = {v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
          if (rp) {
            const f16x4 rr = *reinterpret_cast<const f16x4*>(rp + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += (float)rr[j];
          }
          const f16x4 ov = {(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
          *reinterpret_cast<f16x4*>(yp + 4 * h) = ov;
        }
      }
    }
  }
}

// ================================================================================================================
// Toeplitz-MFMA variant for the wide depthwise kernels (k = 5, 7; f16 mode): 49 taps per output make the stencil VALU-bound
// on the vector units (196 wave-level FMAs per channel per 256 pixels).  Along x a depthwise row filter is a banded matrix:
//     out[c][y][x0+i] = sum_ky sum_k T_c,ky[i][k] * in[c][y+ky][x0+k],     T[i][k] = w[c][ky][k-i]  (0 <= k-i < K)
// i.e. per channel and filter row ONE 16x16x32 MFMA (A = T, 16 outputs x 32 inputs; B = 32 input columns x 16 image rows)
// instead of 16*16*K FMAs: 7 MFMAs per channel per 16x16 tile.  The k index of an MFMA operand must be register-contiguous,
// so the halo tile is transposed to channel-major [c][row][x] on its way into LDS (8 ds_write_b16 per loaded vector); the A
// fragments T_c,ky are precomputed on the host (ey_dsconv_pack_toeplitz) and live in registers: wave w owns channels w, w+8, ...
// The MFMA result (lane = image row, 4 outputs along x) is written as f16 to a pixel-major LDS tile and the pointwise GEMM
// runs from there exactly as in dsconv_kernel.  Persistent 512-thread workgroups, 2 barriers per tile, the next tile's halo
// in flight during the pointwise phase.
#define TZ_RS 24  // halo row stride (elements): 48 B -> the 16 rows a B fragment touches fall into distinct banks
template <int K, int NT, int CPW>
__global__ __launch_bounds__(512) void dsconv_tz_kernel(DsP p, const f16* __restrict__ tz) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef f16 T;
  constexpr int TH = 16, TW = 16, HH = TH + K - 1, HW = TW + K - 1, CPL = HH * TZ_RS;
  constexpr int C = 8 * CPW;                       // channels (8 waves x CPW each)
  constexpr int LSd = C + 8, ROWS = 16 * LSd + 8;  // s_dw: [row][x][LSd], rows padded so that column-of-rows writes spread over banks
  constexpr int NVEC = HH * HW * (C / 8), NHV = (NVEC + 511) / 512;
  T* s_in = reinterpret_cast<T*>(smem);            // [C][HH][TZ_RS]
  T* s_dw = s_in + C * CPL + 32;                   // [16][ROWS]  (+32: B fragments of the last halo row read up to 16 B past a plane)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  for (int i = tid; i < (C * CPL + 32 + 16 * ROWS + 64) / 2; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0u;  // finite everywhere
  // Toeplitz fragments of this wave's channels, pointwise weights, bias -> registers
  Vec8<T> tzf[CPW][K];
#pragma unroll
  for (int cc = 0; cc < CPW; ++cc)
#pragma unroll
    for (int ky = 0; ky < K; ++ky) tzf[cc][ky].load(tz + (((wave + 8 * cc) * K + ky) * 64 + lane) * 8);
  Vec8<T> af[NT];
  {
    const __amdgpu_buffer_rsrc_t rw = ey_rsrc(p.wpw, (unsigned)(16 * NT * p.Kpad * (int)sizeof(T)));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) BufLoad8<T>::load(af[nt], rw, (unsigned)((r * p.Kpad + 8 * g) * (int)sizeof(T)), nt * 16 * p.Kpad * (int)sizeof(T));
  }
  const int ch0 = g * 4 * NT;
  float bias[4 * NT];
#pragma unroll
  for (int i = 0; i < 4 * NT; ++i) bias[i] = (p.bias && ch0 + i < p.Cout) ? p.bias[ch0 + i] : 0.f;
  float dwb[CPW];
#pragma unroll
  for (int cc = 0; cc < CPW; ++cc) dwb[cc] = p.dwbias ? p.dwbias[wave + 8 * cc] : 0.f;
  const __amdgpu_buffer_rsrc_t rx = ey_rsrc(p.x, p.xBytes);
  const int tiles_img = p.tilesX * p.tilesY;
  // this thread's halo vectors (tile independent): channel group cv (slowest, so a wave writes consecutive pixels of one channel), pixel
  int hpo[NHV], hlo[NHV];
#pragma unroll
  for (int u = 0; u < NHV; ++u) {
    const int v = tid + u * 512, cv = v / (HH * HW), px = v - cv * (HH * HW), hy = px / HW, hx = px - hy * HW;
    hpo[u] = v < NVEC ? ((hy << 8) | hx | (cv << 16)) : -1;
    hlo[u] = (8 * cv) * CPL + hy * TZ_RS + hx;
  }
  auto issue_halo = [&](long tile, Vec8<T> (&hv)[NHV]) {
    const int b = (int)(tile / tiles_img), trem = (int)(tile - (long)b * tiles_img);
    const int iy0 = (trem / p.tilesX) * TH - K / 2, ix0 = (trem % p.tilesX) * TW - K / 2;
#pragma unroll
    for (int u = 0; u < NHV; ++u) {
      const int hy = (hpo[u] >> 8) & 0xFF, hx = hpo[u] & 0xFF, cv = hpo[u] >> 16;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = hpo[u] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      BufLoad8<T>::load(hv[u], rx, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * p.xCs + cv * 8) * (int)sizeof(T)) : EY_OOB);
    }
  };
  Vec8<T> hv[NHV];
  // a workgroup owns a contiguous run of tiles, and the runs of an XCD's workgroups are contiguous too: the halo a tile shares with
  // its neighbours is fetched into one L2 (the grid-strided order had every neighbour on another XCD: 1.66-1.81x the algorithmic bytes)
  const long lb = ey_xcd_block(blockIdx.x, gridDim.x), base = p.ntile / gridDim.x, extra = p.ntile - base * gridDim.x;
  long tile = p.xcd ? lb * base + (lb < extra ? lb : extra) : blockIdx.x;
  const long tend = p.xcd ? tile + base + (lb < extra ? 1 : 0) : p.ntile, tstep = p.xcd ? 1 : gridDim.x;
  if (tile < tend) issue_halo(tile, hv);
  __syncthreads();  // LDS zero fill complete
  for (; tile < tend; tile += tstep) {
    // ---- 1. halo -> LDS, transposed to [c][row][x]
#pragma unroll
    for (int u = 0; u < NHV; ++u) {
      if (hpo[u] >= 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s_in[hlo[u] + i * CPL] = hv[u].v[i];
      }
    }
    __syncthreads();
    // ---- 2. depthwise rows as Toeplitz MFMAs; result (lane = image row j, 4 outputs x = 4g..4g+3) -> s_dw[row][x][c]
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
      const int c = wave + 8 * cc;
      f32x4 acc = (f32x4)0.f;
      const T* bp = s_in + c * CPL + r * TZ_RS + 8 * g;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        Vec8<T> bfr;
        bfr.load(bp + ky * TZ_RS);
        acc = ds_mma16(tzf[cc][ky], bfr, acc);
      }
      T* dp = s_dw + r * ROWS + (4 * g) * LSd + c;
      float dv[4] = {acc[0] + dwb[cc], acc[1] + dwb[cc], acc[2] + dwb[cc], acc[3] + dwb[cc]};
      ey_act_n(dv, p.dwact);
#pragma unroll
      for (int t = 0; t < 4; ++t) dp[t * LSd] = (T)dv[t];
    }
    __syncthreads();
    const long cur = tile, nxt = tile + tstep;
    if (nxt < tend) issue_halo(nxt, hv);  // in flight during the pointwise phase
    // ---- 3. pointwise GEMM: wave handles image rows wave and wave+8 of the tile (16 pixels each)
    const int b = (int)(cur / tiles_img), trem = (int)(cur - (long)b * tiles_img);
    const int ty0 = (trem / p.tilesX) * TH, tx0 = (trem % p.tilesX) * TW;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = wave + 8 * h;
      Vec8<T> bfr;
      bfr.load(s_dw + row * ROWS + r * LSd + 8 * g);  // lanes with 8g >= C read pad / neighbour data against zero weight slack
      f32x4 pacc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) pacc[nt] = ds_mma16(af[nt], bfr, (f32x4)0.f);
      const int oy = ty0 + row, ox = tx0 + r;
      if (oy >= p.H || ox >= p.W) continue;
      const long m = ((long)b * p.H + oy) * p.W + ox;
      float v[4 * NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * nt + j] = pacc[nt][j] + bias[4 * nt + j];
      ey_act_n(v, p.act);
      T* yp = (T*)p.y + m * p.yCs + ch0;
      const T* rp = p.res ? (const T*)p.res + m * p.resCs + ch0 : nullptr;
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        if (ch0 + 4 * q + 4 <= p.Cout) {
          float o[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
          if (rp) {
            const f16x4 rr = *reinterpret_cast<const f16x4*>(rp + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += (float)rr[j];
          }
          const f16x4 ov = {(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
          *reinterpret_cast<f16x4*>(yp + 4 * q) = ov;
        }
      }
    }
  }
}

// mirrors conv_igemm.hip (same packing rule)
static int ds_conv_nt(int Cout) {
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

template <typename T, int K, int NT>
static int ds_launch(DsP p, hipStream_t st) {
  const int C = p.Cin, HH = DS_TH + K - 1, HW = DS_TW + K - 1;
  const int LSw = p.Kpad + (((p.Kpad >> 3) & 1) ? 0 : 8);
  const size_t lds = ((size_t)HH * HW * C + (size_t)K * K * C) * 4 + ((size_t)DS_TH * DS_TW * (C + 8) + (size_t)16 * NT * LSw) * sizeof(T);
  if (lds > 160 * 1024) return ey_set_error(EY_EUNSUPPORTED, "dsconv: Cin=%d k=%d needs %zu B of LDS", C, K, lds);
  static size_t reserved = 0;
  if (lds > 64 * 1024 && lds > reserved) {
    if (hipFuncSetAttribute((const void*)dsconv_kernel<T, K, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "dsconv: cannot reserve %zu B of LDS", lds);
    reserved = lds;
  }
  const int ntn = (p.Cout + 16 * NT - 1) / (16 * NT);
  static size_t occ_lds = ~(size_t)0;
  static int occ = 1;
  if (occ_lds != lds) {  // resident workgroups per CU the LDS and the registers allow (the grid is persistent)
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)dsconv_kernel<T, K, NT>, 256, lds) != hipSuccess || n < 1) n = 1;
    occ = n;
    occ_lds = lds;
  }
  int per_cu = occ > 6 ? 6 : occ;
  long gx = (long)256 * per_cu / ntn;
  if (gx < 1) gx = 1;
  if (gx > p.ntile) gx = p.ntile;
  g_ds_variant = 1;
  hipLaunchKernelGGL((dsconv_kernel<T, K, NT>), dim3((unsigned)gx, ntn), dim3(256), lds, st, p);
  EY_LAUNCH_CHECK("ey_dsconv");
  return EY_OK;
}

template <typename T, int K>
static int ds_launch_nt(const DsP& p, hipStream_t st) {
  switch (p.NTpack) {
    case 1: return ds_launch<T, K, 1>(p, st);
    case 2: return ds_launch<T, K, 2>(p, st);
    case 4: return ds_launch<T, K, 4>(p, st);
    case 5: return ds_launch<T, K, 5>(p, st);
    default: return ds_launch<T, K, 8>(p, st);
  }
}

template <typename T>
static int ds_launch_k(const DsP& p, int k, hipStream_t st) {
  if (k == 3) return ds_launch_nt<T, 3>(p, st);
  if (k == 5) return ds_launch_nt<T, 5>(p, st);
  return ds_launch_nt<T, 7>(p, st);
}

// ---- register-strip dispatch: f16, Cout <= 64 covered by ONE channel tile (NT = NTpack), Cin <= 64, 4-element vector stores
#include <stdlib.h>
// strip length P: long strips reuse every loaded vector for more outputs, short strips give small maps enough waves
// (a 20x20x32 map is 800 waves of 16 pixels; the per-wave instruction stream, not bandwidth, sets the time there)
template <int K, int NT, int KS, int P>
static int ds_strip_launch(const DsP& p, hipStream_t st) {
  const long nstrip = (long)p.B * p.H * ((p.W + P - 1) / P);
  const long nwave = (nstrip + 15) / 16;
  const size_t lds = (size_t)K * K * p.Cin * sizeof(f16);
  g_ds_variant = 2;
  hipLaunchKernelGGL((dsconv_strip_kernel<K, NT, KS, P>), dim3((unsigned)((nwave + 3) / 4)), dim3(256), lds, st, p);
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) return ey_set_error(EY_ELAUNCH, "ey_dsconv(strip): %s", hipGetErrorString(e_));
  return 1;
}
template <int K, int NT, int KS>
static int ds_strip_p(const DsP& p, hipStream_t st) {
  // measured on MI355X (tools/ds_bench.py): k=7 wants the long strip (49 taps: reuse of loaded vectors dominates), k=3/5 the
  // short ones (more waves; the per-wave instruction stream sets the time), 2-pixel strips once a 64-channel map has >= 40k pixels
  const long force = tune().ds_p;
  const long px = (long)p.B * p.H * p.W;
  int P = K == 7 ? 4 : (p.Cin >= 64 && px >= 40000) ? 2 : 1;
  if (force) P = (int)force;
  if (P == 4) return ds_strip_launch<K, NT, KS, 4>(p, st);
  if (P == 2) return ds_strip_launch<K, NT, KS, 2>(p, st);
  return ds_strip_launch<K, NT, KS, 1>(p, st);
}
template <int K>
static int ds_strip_k(const DsP& p, int nt, int ks, hipStream_t st) {
#define DSS(NTV, KSV) if (nt == NTV && ks == KSV) return ds_strip_p<K, NTV, KSV>(p, st);
  DSS(1, 1) DSS(2, 1) DSS(4, 2)
#undef DSS
  return 0;
}
static int ds_strip_dispatch(const DsP& p, int k, hipStream_t st) {
  if (!tune().ds_strip) return 0;
  const int nt = p.NTpack, ks = (p.Cin + 31) / 32;
  if (p.Cout > 16 * nt || ks > 2 || !p.vec_store || p.Cout % 4) return 0;
  if (p.Cin < 32) return 0;  // 16 channels would leave half of every wave's lanes (channel groups 2,3) idle: the LDS-tile kernel wins
  if ((long)p.B * p.H * p.W >= (1L << 29)) return 0;
  if (k == 3) return ds_strip_k<3>(p, nt, ks, st);
  if (k == 5) return ds_strip_k<5>(p, nt, ks, st);
  return ds_strip_k<7>(p, nt, ks, st);
}

extern "C" int ey_dsconv_last_variant(void) { return g_ds_variant; }

// ---- Toeplitz fragments: out[((c*k + ky)*64 + lane)*8 + t] = w[ky][kx = 8g + t - i][c], lane = (i = lane & 15, g = lane >> 4)
extern "C" size_t ey_dsconv_toeplitz_bytes(int C, int k) { return (size_t)C * k * 64 * 8 * sizeof(f16); }
extern "C" int ey_dsconv_pack_toeplitz(int C, int k, const float* w_kkc_host, void* out_host, size_t out_bytes) {
  EY_CHECK(C > 0 && (k == 3 || k == 5 || k == 7) && w_kkc_host && out_host, "pack_toeplitz: bad arguments");
  EY_CHECK(out_bytes >= ey_dsconv_toeplitz_bytes(C, k), "pack_toeplitz: output buffer too small");
  f16* o = (f16*)out_host;
  for (int c = 0; c < C; ++c)
    for (int ky = 0; ky < k; ++ky)
      for (int lane = 0; lane < 64; ++lane)
        for (int t = 0; t < 8; ++t) {
          const int i = lane & 15, g = lane >> 4, kx = 8 * g + t - i;
          o[(((size_t)c * k + ky) * 64 + lane) * 8 + t] = (kx >= 0 && kx < k) ? (f16)w_kkc_host[((size_t)ky * k + kx) * C + c] : (f16)0.f;
        }
  return EY_OK;
}

template <int K, int NT, int CPW>
static int ds_tz_launch(DsP p, const void* tz, hipStream_t st) {
  constexpr int C = 8 * CPW, HH = 16 + K - 1, LSd = C + 8, ROWS = 16 * LSd + 8;
  const size_t lds = ((size_t)C * HH * TZ_RS + 32 + 16 * ROWS + 64) * sizeof(f16);
  p.tilesX = (p.W + 15) / 16; p.tilesY = (p.H + 15) / 16;
  p.ntile = (long)p.B * p.tilesX * p.tilesY;
  static int occ = 0;
  if (!occ) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)dsconv_tz_kernel<K, NT, CPW>, 512, lds) != hipSuccess || n < 1) n = 1;
    occ = n > 2 ? 2 : n;
  }
  long gx = 256L * occ;
  if (gx > p.ntile) gx = p.ntile;
  g_ds_variant = 3;
  hipLaunchKernelGGL((dsconv_tz_kernel<K, NT, CPW>), dim3((unsigned)gx), dim3(512), lds, st, p, (const f16*)tz);
  EY_LAUNCH_CHECK("ey_dsconv_tz");
  return EY_OK;
}

extern "C" int ey_dsconv(int dtype, int B, int H, int W, int Cin, int Cout, int k, int act, const void* x, int x_cstride, const void* w_dw_kkc,
                         const float* dw_bias, int dw_act, const void* w_pw_packed, const float* bias, void* y, int y_cstride, const void* res, int res_cstride, ey_stream_t stream);

extern "C" int ey_dsconv_tz(int dtype, int B, int H, int W, int Cin, int Cout, int k, int act, const void* x, int x_cstride, const void* w_dw_kkc,
                            const void* w_dw_toeplitz, const float* dw_bias, int dw_act, const void* w_pw_packed, const float* bias, void* y, int y_cstride,
                            const void* res, int res_cstride, ey_stream_t stream) {
  const int es = 2;
  // measured (tools/ds_bench.py): k = 7 always wins (85 -> 28 us at C16 160x160); k = 3/5 win once the map has >= 100k pixels
  // (33 -> 24 us), below that the register-strip kernel's extra waves matter more.  EY_TZ_KMASK (decimal bit mask over k) overrides.
  const long tz_kmask = tune().tz_kmask, tz_minpx = tune().tz_minpx;
  const bool fits = dtype == EY_F16 && w_dw_toeplitz && (k == 3 || k == 5 || k == 7) && ((tz_kmask >> k) & 1) && (k == 7 || (long)B * H * W >= tz_minpx) &&
                    (Cin == 16 || Cin == 32) && Cout <= 32 && Cout % 4 == 0 && x && y && w_pw_packed &&
                    x_cstride >= Cin && (x_cstride * es) % 16 == 0 && ey_aligned(x, 16) && ey_aligned(w_dw_toeplitz, 16) && (y_cstride * es) % 8 == 0 && ey_aligned(y, 8) &&
                    (!res || ((res_cstride * es) % 8 == 0 && ey_aligned(res, 8))) && y_cstride >= Cout && (((long)B * H * W - 1) * x_cstride + Cin) * es < (1L << 31);
  if (!fits)  // every other shape: the general entry point
    return ey_dsconv(dtype, B, H, W, Cin, Cout, k, act, x, x_cstride, w_dw_kkc, dw_bias, dw_act, w_pw_packed, bias, y, y_cstride, res, res_cstride, stream);
  DsP p;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.act = act;
  p.x = x; p.xCs = x_cstride; p.xBytes = (unsigned)((((long)B * H * W - 1) * x_cstride + Cin) * es);
  p.wdw = w_dw_kkc; p.dwbias = dw_bias; p.dwact = dw_act; p.wpw = w_pw_packed; p.bias = bias; p.y = y; p.yCs = y_cstride; p.res = res; p.resCs = res_cstride;
  p.Kpad = ey_conv_kpad(Cin, es);
  p.NTpack = ds_conv_nt(Cout);
  p.vec_store = 1;
  p.xcd = (int)((tune().xcd_map >> 1) & 1);
  hipStream_t st = (hipStream_t)stream;
#define TZ(KV)                                                                                                                             \
  if (k == KV) {                                                                                                                       \
    if (Cin == 16) return p.NTpack == 1 ? ds_tz_launch<KV, 1, 2>(p, w_dw_toeplitz, st) : ds_tz_launch<KV, 2, 2>(p, w_dw_toeplitz, st); \
    return p.NTpack == 1 ? ds_tz_launch<KV, 1, 4>(p, w_dw_toeplitz, st) : ds_tz_launch<KV, 2, 4>(p, w_dw_toeplitz, st);               \
  }
  TZ(3) TZ(5) TZ(7)
#undef TZ
  return ey_set_error(EY_EINVAL, "dsconv_tz: k=%d", k);
}

extern "C" int ey_dsconv(int dtype, int B, int H, int W, int Cin, int Cout, int k, int act, const void* x, int x_cstride, const void* w_dw_kkc,
                         const float* dw_bias, int dw_act, const void* w_pw_packed, const float* bias, void* y, int y_cstride, const void* res, int res_cstride, ey_stream_t stream) {
  EY_CHECK(x && w_dw_kkc && w_pw_packed && y, "dsconv: null pointer");
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "dsconv: bad dtype");
  EY_CHECK(k == 3 || k == 5 || k == 7, "dsconv: k=%d (3,5,7)", k);
  EY_CHECK(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "dsconv: bad extent");
  EY_CHECK(Cin % 8 == 0 && Cin <= 256, "dsconv: Cin=%d must be a multiple of 8 and <= 256", Cin);
  const int es = dtype == EY_F16 ? 2 : 4;
  EY_CHECK(x_cstride >= Cin && (x_cstride * es) % 16 == 0 && ey_aligned(x, 16) && ey_aligned(w_dw_kkc, 16) && ey_aligned(w_pw_packed, 16),
           "dsconv: input view / weights must be 16-byte aligned");
  EY_CHECK(y_cstride >= Cout && (!res || res_cstride >= Cout), "dsconv: cstride");
  EY_CHECK(!dw_bias || ey_aligned(dw_bias, 16), "dsconv: dw_bias must be 16-byte aligned");
  const long xbytes = (((long)B * H * W - 1) * x_cstride + Cin) * es;
  if (xbytes >= (1L << 31)) return ey_set_error(EY_EUNSUPPORTED, "dsconv: input view larger than 2 GiB");
  DsP p;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.act = act;
  p.x = x; p.xCs = x_cstride; p.xBytes = (unsigned)xbytes;
  p.wdw = w_dw_kkc; p.dwbias = dw_bias; p.dwact = dw_act; p.wpw = w_pw_packed; p.bias = bias; p.y = y; p.yCs = y_cstride; p.res = res; p.resCs = res_cstride;
  p.Kpad = ey_conv_kpad(Cin, es);
  p.NTpack = ds_conv_nt(Cout);
  p.tilesX = (W + DS_TW - 1) / DS_TW; p.tilesY = (H + DS_TH - 1) / DS_TH;
  p.ntile = (long)B * p.tilesX * p.tilesY;
  p.xcd = 0;
  const int va = 4 * es;
  p.vec_store = Cout % 4 == 0 && (y_cstride * es) % va == 0 && ey_aligned(y, va) && (!res || ((res_cstride * es) % va == 0 && ey_aligned(res, va)));
  if (p.vec_store && (y_cstride * es) % 16 == 0 && ey_aligned(y, 16) && (!res || ((res_cstride * es) % 16 == 0 && ey_aligned(res, 16)))) p.vec_store = 2;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EY_F16) {
    const int rs = ds_strip_dispatch(p, k, st);
    if (rs != 0) return rs < 0 ? rs : EY_OK;
  }
  return dtype == EY_F16 ? ds_launch_k<f16>(p, k, st) : ds_launch_k<float>(p, k, st);
}
