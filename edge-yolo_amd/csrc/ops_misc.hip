// HBM-bound NHWC operators of the EdgeLine-YOLO forward path: stem conv, depthwise conv, Haar DWT, SPPF pooling,
// slice copies / layout transposes, and the generic scalar direct convolution (correctness path for odd shapes).
#include "common.h"
#include "tune.h"
#include <stdlib.h>

// ============================================================================ generic direct conv (scalar)
template <typename T>
__global__ void conv_direct_kernel(ey_conv_direct_desc d) {
  const long total = (long)d.B * d.Ho * d.Wo * d.Cout;
  const int cpg_in = d.Cin / d.groups, cpg_out = d.Cout / d.groups;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int co = (int)(idx % d.Cout);
    long m = idx / d.Cout;
    const int ox = (int)(m % d.Wo);
    m /= d.Wo;
    const int oy = (int)(m % d.Ho);
    const int b = (int)(m / d.Ho);
    const int grp = co / cpg_out;
    float acc = d.bias ? d.bias[co] : 0.f;
    for (int ky = 0; ky < d.k; ++ky) {
      const int iy = oy * d.stride - d.pad + ky;
      if (iy < 0 || iy >= d.H) continue;
      for (int kx = 0; kx < d.k; ++kx) {
        const int ix = ox * d.stride - d.pad + kx;
        if (ix < 0 || ix >= d.W) continue;
        const T* xp = (const T*)d.x + (((long)b * d.H + iy) * d.W + ix) * d.x_cstride + grp * cpg_in;
        const float* wp = d.w_oihw + ((long)co * cpg_in * d.k + ky) * d.k + kx;
        for (int c = 0; c < cpg_in; ++c) acc += to_f(xp[c]) * wp[(long)c * d.k * d.k];
      }
    }
    ((T*)d.y)[(((long)b * d.Ho + oy) * d.Wo + ox) * d.y_cstride + co] = from_f<T>(ey_act(acc, d.act));
  }
}

extern "C" int ey_conv2d_direct(const ey_conv_direct_desc* d, ey_stream_t stream) {
  EY_CHECK(d && d->x && d->w_oihw && d->y, "conv_direct: null pointer");
  EY_CHECK(d->dtype == EY_F16 || d->dtype == EY_F32, "conv_direct: bad dtype");
  EY_CHECK(d->groups > 0 && d->Cin % d->groups == 0 && d->Cout % d->groups == 0, "conv_direct: groups=%d Cin=%d Cout=%d", d->groups, d->Cin, d->Cout);
  EY_CHECK(d->k > 0 && d->stride > 0 && d->Ho == (d->H + 2 * d->pad - d->k) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->k) / d->stride + 1,
           "conv_direct: inconsistent extents");
  EY_CHECK(d->x_cstride >= d->Cin && d->y_cstride >= d->Cout, "conv_direct: cstride");
  const long total = (long)d->B * d->Ho * d->Wo * d->Cout;
  const int blocks = (int)((total + 255) / 256 < 65535 * 16 ? (total + 255) / 256 : 65535 * 16);
  if (d->dtype == EY_F16) hipLaunchKernelGGL(conv_direct_kernel<f16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL(conv_direct_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d);
  EY_LAUNCH_CHECK("ey_conv2d_direct");
  return EY_OK;
}

// ============================================================================ stem: NCHW image -> NHWC, 3x3 s2
// One thread = TWO horizontally adjacent output pixels x 16 output channels: the 5 x 3 input window per channel is
// read once for both (range-checked buffer loads: the zero padding and the right/bottom edges need no branches; f16
// columns are fetched as aligned pairs), the 2 x 16 results leave as 64 contiguous bytes.  Weights are wave-uniform.
template <typename TI> struct StemLoad;
template <> struct StemLoad<f16> {
  // columns (c0-1, c0, c0+1, c0+2, c0+3) of one image row; c0 even
  static __device__ __forceinline__ void row5(__amdgpu_buffer_rsrc_t r, long rowbase, int c0, int W, bool rowok, float (&o)[5]) {
    const unsigned b0 = (unsigned)((rowbase + c0) * 2);
    const unsigned short s = __builtin_amdgcn_raw_buffer_load_b16(r, (rowok && c0 >= 1) ? b0 - 2u : EY_OOB, 0, 0);
    const unsigned p0 = __builtin_amdgcn_raw_buffer_load_b32(r, (rowok && c0 + 1 < W) ? b0 : EY_OOB, 0, 0);
    const unsigned p1 = __builtin_amdgcn_raw_buffer_load_b32(r, (rowok && c0 + 3 < W) ? b0 + 4u : EY_OOB, 0, 0);
    o[0] = (float)__builtin_bit_cast(f16, s);
    o[1] = (float)__builtin_bit_cast(f16, (unsigned short)(p0 & 0xFFFFu));
    o[2] = (float)__builtin_bit_cast(f16, (unsigned short)(p0 >> 16));
    o[3] = (float)__builtin_bit_cast(f16, (unsigned short)(p1 & 0xFFFFu));
    o[4] = (float)__builtin_bit_cast(f16, (unsigned short)(p1 >> 16));
  }
};
template <> struct StemLoad<float> {
  static __device__ __forceinline__ void row5(__amdgpu_buffer_rsrc_t r, long rowbase, int c0, int W, bool rowok, float (&o)[5]) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = c0 - 1 + i;
      o[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (rowok && c >= 0 && c < W) ? (unsigned)((rowbase + c) * 4) : EY_OOB, 0, 0));
    }
  }
};

template <typename TI, typename TO, int CIN>
__global__ __launch_bounds__(256) void stem_kernel(int B, int H, int W, int Ho, int Wo, int Cout, int act, const TI* __restrict__ x, unsigned xbytes,
                                                   const float* __restrict__ w, const float* __restrict__ bias, TO* __restrict__ y, int yCs) {
  const int Wp = (Wo + 1) >> 1;  // pixel pairs per output row
  const long npair = (long)B * Ho * Wp;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npair) return;
  const int co0 = blockIdx.y * 16;
  const int px = (int)(t % Wp);
  const long tt = t / Wp;
  const int oy = (int)(tt % Ho), b = (int)(tt / Ho);
  const int ox = px * 2, c0 = ox * 2;  // input column of tap kx=1 of the first pixel
  const __amdgpu_buffer_rsrc_t rs = ey_rsrc(x, xbytes);
  float in[CIN][3][5];
#pragma unroll
  for (int c = 0; c < CIN; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      StemLoad<TI>::row5(rs, (((long)b * CIN + c) * H + iy) * W, c0, W, iy >= 0 && iy < H, in[c][ky]);
    }
  float a0[16], a1[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    float s0 = bias ? bias[co0 + o] : 0.f, s1 = s0;
    const float* wo = w + (long)(co0 + o) * CIN * 9;
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float wv = wo[(c * 3 + ky) * 3 + kx];
          s0 += in[c][ky][kx] * wv;
          s1 += in[c][ky][kx + 2] * wv;
        }
    a0[o] = ey_act(s0, act);
    a1[o] = ey_act(s1, act);
  }
  TO* yp = y + (((long)b * Ho + oy) * Wo + ox) * yCs + co0;
  Vec8<TO> v;
#pragma unroll
  for (int o = 0; o < 8; ++o) v.set(o, a0[o]);
  v.store(yp);
#pragma unroll
  for (int o = 0; o < 8; ++o) v.set(o, a0[8 + o]);
  v.store(yp + 8);
  if (ox + 1 < Wo) {
#pragma unroll
    for (int o = 0; o < 8; ++o) v.set(o, a1[o]);
    v.store(yp + yCs);
#pragma unroll
    for (int o = 0; o < 8; ++o) v.set(o, a1[8 + o]);
    v.store(yp + yCs + 8);
  }
}

// MFMA stem (f16 image -> f16 NHWC, Cin = 3, W % 8 == 0): the direct kernel above spends 27 x Cout FMAs per pixel on the
// vector units (864 per thread) and is VALU-bound at 2.4 TB/s.  As a GEMM the layer is K = 27 (padded to 32), N = Cout: ONE
// 16x16x32 MFMA per 16 pixels per 16 channels.  A workgroup stages the 3 x 17 x 144 input patch of an 8 x 64 output tile into LDS
// with aligned 16-byte loads of the planar image (range-checked: zero padding), then every lane gathers its 8 taps
// (k = c*9 + ky*3 + kx, the OIHW order) with 2-byte LDS reads; weights are converted once per wave into the A fragment.
#define STM_TR 8
#define STM_TC 64
#define STM_LW 152  // LDS row stride (elements): 144 loaded columns + 8
template <int NT>
__global__ __launch_bounds__(256) void stem_mfma_kernel(int B, int H, int W, int Ho, int Wo, int act, const f16* __restrict__ x, unsigned xbytes,
                                                        const float* __restrict__ w, const float* __restrict__ bias, f16* __restrict__ y, int yCs) {
  constexpr int HR = 2 * STM_TR + 1, NV = 3 * HR * 18;
  __shared__ __attribute__((aligned(16))) f16 s_in[3 * HR * STM_LW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int tilesX = (Wo + STM_TC - 1) / STM_TC, tilesY = (Ho + STM_TR - 1) / STM_TR;
  const int b = blockIdx.x / (tilesX * tilesY), trem = blockIdx.x - b * (tilesX * tilesY);
  const int oy0 = (trem / tilesX) * STM_TR, ox0 = (trem % tilesX) * STM_TC;
  const int iy0 = 2 * oy0 - 1, ixa = 2 * ox0 - 8;  // first staged input row / (8-aligned) column
  const __amdgpu_buffer_rsrc_t rs = ey_rsrc(x, xbytes);
#pragma unroll
  for (int u = 0; u < (NV + 255) / 256; ++u) {
    const int v = tid + u * 256;
    if (v < NV) {
      const int c = v / (HR * 18), rem = v - c * (HR * 18), row = rem / 18, vc = rem - row * 18;
      const int iy = iy0 + row, ix = ixa + 8 * vc;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;  // W % 8 == 0: a vector is entirely inside or entirely outside
      Vec8<f16> t;
      BufLoad8<f16>::load(t, rs, ok ? (unsigned)(((((long)b * 3 + c) * H + iy) * W + ix) * 2) : EY_OOB);
      t.store(s_in + (c * HR + row) * STM_LW + 8 * vc);
    }
  }
  // A fragments (weights, fp32 OIHW -> f16) and this lane's tap offsets: k = 8g + t = c*9 + ky*3 + kx
  Vec8<f16> af[NT];
  int off[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int k = 8 * g + t, c = k / 9, ky = (k - 9 * c) / 3, kx = k - 9 * c - 3 * ky;
    off[t] = k < 27 ? (c * HR + ky) * STM_LW + kx : 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) af[nt].v[t] = k < 27 ? (f16)w[(nt * 16 + r) * 27 + k] : (f16)0.f;
  }
  float bs[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[nt][j] = bias ? bias[nt * 16 + 4 * g + j] : 0.f;
  __syncthreads();
  // wave w: output rows 2w, 2w+1 of the tile, 4 column blocks of 16 pixels each
#pragma unroll
  for (int blk = 0; blk < 8; ++blk) {
    const int rr = 2 * wave + (blk >> 2), j = (blk & 3) * 16 + r;
    const f16* bp = s_in + (2 * rr) * STM_LW + 2 * j + 7;
    Vec8<f16> bq;
#pragma unroll
    for (int t = 0; t < 8; ++t) bq.v[t] = bp[off[t]];
    const int oy = oy0 + rr, ox = ox0 + j;
    const bool ok = oy < Ho && ox < Wo;
    f16* yp = y + (((long)b * Ho + oy) * Wo + ox) * yCs + 4 * g;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt].v, bq.v, (f32x4)0.f, 0, 0, 0);
      if (ok) {
        float v4[4] = {acc[0] + bs[nt][0], acc[1] + bs[nt][1], acc[2] + bs[nt][2], acc[3] + bs[nt][3]};
        ey_act_n(v4, act);
        const f16x4 o = {(f16)v4[0], (f16)v4[1], (f16)v4[2], (f16)v4[3]};
        *reinterpret_cast<f16x4*>(yp + nt * 16) = o;
      }
    }
  }
}

template <int NT>
static int stem_mfma_launch(int B, int H, int W, int act, const void* x, const float* w, const float* bias, void* y, int yCs, hipStream_t st) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long tiles = (long)B * ((Wo + STM_TC - 1) / STM_TC) * ((Ho + STM_TR - 1) / STM_TR);
  hipLaunchKernelGGL((stem_mfma_kernel<NT>), dim3((unsigned)tiles), dim3(256), 0, st, B, H, W, Ho, Wo, act, (const f16*)x, (unsigned)((long)B * 3 * H * W * 2), w, bias,
                     (f16*)y, yCs);
  EY_LAUNCH_CHECK("ey_stem_conv(mfma)");
  return EY_OK;
}

template <typename TI, typename TO>
static int stem_launch(int B, int Cin, int H, int W, int Cout, int act, const void* x, const float* w, const float* bias, void* y, int yCs, hipStream_t st) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long npair = (long)B * Ho * ((Wo + 1) / 2);
  const long xbytes = (long)B * Cin * H * W * (long)sizeof(TI);
  if (xbytes >= (1L << 31)) return ey_set_error(EY_EUNSUPPORTED, "stem: input batch larger than 2 GiB");
  dim3 grid((unsigned)((npair + 255) / 256), Cout / 16);
#define STEM(CI) hipLaunchKernelGGL((stem_kernel<TI, TO, CI>), grid, dim3(256), 0, st, B, H, W, Ho, Wo, Cout, act, (const TI*)x, (unsigned)xbytes, w, bias, (TO*)y, yCs)
  switch (Cin) {
    case 1: STEM(1); break;
    case 2: STEM(2); break;
    case 3: STEM(3); break;
    default: STEM(4); break;
  }
#undef STEM
  EY_LAUNCH_CHECK("ey_stem_conv");
  return EY_OK;
}

extern "C" int ey_stem_conv(int x_dtype, int y_dtype, int B, int Cin, int H, int W, int Cout, int act, const void* x, const float* w,
                            const float* bias, void* y, int y_cstride, ey_stream_t stream) {
  EY_CHECK(x && w && y, "stem: null pointer");
  EY_CHECK(Cin >= 1 && Cin <= 4 && Cout % 16 == 0 && Cout > 0, "stem: Cin=%d (1..4) Cout=%d (multiple of 16)", Cin, Cout);
  EY_CHECK(B > 0 && H > 0 && W > 0, "stem: bad extent");
  const int es = y_dtype == EY_F16 ? 2 : 4;
  EY_CHECK(y_cstride >= Cout && (y_cstride * es) % 16 == 0 && ey_aligned(y, 16), "stem: output view not 16-byte aligned");
  EY_CHECK(x_dtype != EY_F16 || (W % 2 == 0 && ey_aligned(x, 4)), "stem: f16 images need an even width (column pairs are fetched as 32-bit words)");
  hipStream_t st = (hipStream_t)stream;
  const bool stem_mfma_off = !tune().stem_mfma;
  if (x_dtype == EY_F16 && y_dtype == EY_F16 && Cin == 3 && W % 8 == 0 && Cout <= 64 && ey_aligned(x, 16) && !stem_mfma_off && (long)B * 3 * H * W * 2 < (1L << 31)) {
    switch (Cout / 16) {  // fp32-accumulated f16 products; the weights are rounded to f16 like every other conv of the f16 mode
      case 1: return stem_mfma_launch<1>(B, H, W, act, x, w, bias, y, y_cstride, st);
      case 2: return stem_mfma_launch<2>(B, H, W, act, x, w, bias, y, y_cstride, st);
      case 3: return stem_mfma_launch<3>(B, H, W, act, x, w, bias, y, y_cstride, st);
      case 4: return stem_mfma_launch<4>(B, H, W, act, x, w, bias, y, y_cstride, st);
    }
  }
  if (x_dtype == EY_F16 && y_dtype == EY_F16) return stem_launch<f16, f16>(B, Cin, H, W, Cout, act, x, w, bias, y, y_cstride, st);
  if (x_dtype == EY_F32 && y_dtype == EY_F16) return stem_launch<float, f16>(B, Cin, H, W, Cout, act, x, w, bias, y, y_cstride, st);
  if (x_dtype == EY_F16 && y_dtype == EY_F32) return stem_launch<f16, float>(B, Cin, H, W, Cout, act, x, w, bias, y, y_cstride, st);
  if (x_dtype == EY_F32 && y_dtype == EY_F32) return stem_launch<float, float>(B, Cin, H, W, Cout, act, x, w, bias, y, y_cstride, st);
  return ey_set_error(EY_EINVAL, "stem: bad dtype");
}

// ============================================================================ depthwise k x k, stride 1
// thread = (pixel, 8-channel vector); taps come through L1/L2 (activation tiles of neighbouring threads overlap).
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_kernel(int B, int H, int W, int C, int act, const T* __restrict__ x, int xCs,
                                                     const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, int yCs) {
  const int cv = C >> 3;
  const long total = (long)B * H * W * cv;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c8 = (int)(idx % cv) * 8;
  long m = idx / cv;
  const int ox = (int)(m % W);
  const long t = m / W;
  const int oy = (int)(t % H);
  const int b = (int)(t / H);
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = bias ? bias[c8 + i] : 0.f;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int iy = oy - K / 2 + ky;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int ix = ox - K / 2 + kx;
      if (ix < 0 || ix >= W) continue;
      Vec8<T> xv, wv;
      xv.load(x + (((long)b * H + iy) * W + ix) * xCs + c8);
      wv.load(w + (ky * K + kx) * C + c8);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += xv.get(i) * wv.get(i);
    }
  }
  Vec8<T> o;
  ey_act_n(acc, act);
#pragma unroll
  for (int i = 0; i < 8; ++i) o.set(i, acc[i]);
  o.store(y + m * yCs + c8);
}

// 3x3 strip variant (the DWConv of the detection heads): a thread owns P consecutive output pixels x 8 channels.  The
// 3 x (P+2) input vectors come by range-checked buffer loads (zero padding without branches, all in flight at once), the 9
// weight vectors and the bias live in registers, and every loaded vector feeds up to 3 outputs per row: 18 loads and
// (for f16) 288 v_fma_mix per 4 outputs instead of 36 loads + per-tap address arithmetic + 72 conversions per output.
template <typename T, int P>
__global__ __launch_bounds__(256) void dwconv3_strip_kernel(int B, int H, int W, int C, int act, int xcd, const T* __restrict__ x, int xCs, unsigned xBytes,
                                                            const T* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, int yCs) {
  const int cv = C >> 3, WS = (W + P - 1) / P;
  const int idx = (int)(xcd ? ey_xcd_block(blockIdx.x, gridDim.x) : blockIdx.x) * 256 + threadIdx.x;  // neighbouring rows' strips in one XCD's L2
  if (idx >= B * H * WS * cv) return;
  const int c8 = (idx % cv) * 8;
  int t = idx / cv;
  const int x0 = (t % WS) * P;
  t /= WS;
  const int oy = t % H, b = t / H;
  const __amdgpu_buffer_rsrc_t rx = ey_rsrc(x, xBytes);
  Vec8<T> xin[3][P + 2];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy - 1 + ky;
    const bool yok = iy >= 0 && iy < H;
    const int rowoff = ((b * H + iy) * W + x0 - 1) * xCs + c8;
#pragma unroll
    for (int j = 0; j < P + 2; ++j) {
      const int ix = x0 - 1 + j;
      BufLoad8<T>::load(xin[ky][j], rx, (yok && ix >= 0 && ix < W) ? (unsigned)((rowoff + j * xCs) * (int)sizeof(T)) : EY_OOB);
    }
  }
  Vec8<T> wv[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) wv[tp].load(w + tp * C + c8);
  float acc[P][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float bb = bias ? bias[c8 + i] : 0.f;
#pragma unroll
    for (int q = 0; q < P; ++q) acc[q][i] = bb;
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int q = 0; q < P; ++q) ey_fma8_mix(xin[ky][q + kx], wv[ky * 3 + kx], acc[q]);
  T* yp = y + (long)((b * H + oy) * W + x0) * yCs + c8;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    if (x0 + q < W) {
      Vec8<T> o;
      ey_act_n(acc[q], act);
#pragma unroll
      for (int i = 0; i < 8; ++i) o.set(i, acc[q][i]);
      o.store(yp + (long)q * yCs);
    }
  }
}

template <typename T>
static int dw_launch(int B, int H, int W, int C, int k, int act, const void* x, int xCs, const void* w, const float* bias, void* y, int yCs, hipStream_t st) {
  const long total = (long)B * H * W * (C / 8);
  const long xbytes = (((long)B * H * W - 1) * xCs + C) * (long)sizeof(T);
  if (k == 3 && xbytes < (1L << 31) && total < (1L << 31)) {
    constexpr int P = sizeof(T) == 2 ? 4 : 2;
    const long n = (long)B * H * ((W + P - 1) / P) * (C / 8);
    hipLaunchKernelGGL((dwconv3_strip_kernel<T, P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B, H, W, C, act, (int)(tune().xcd_map & 1), (const T*)x, xCs, (unsigned)xbytes,
                       (const T*)w, bias, (T*)y, yCs);
    EY_LAUNCH_CHECK("ey_dwconv");
    return EY_OK;
  }
  dim3 grid((unsigned)((total + 255) / 256));
#define DW(K) hipLaunchKernelGGL((dwconv_kernel<T, K>), grid, dim3(256), 0, st, B, H, W, C, act, (const T*)x, xCs, (const T*)w, bias, (T*)y, yCs)
  if (k == 3) DW(3); else if (k == 5) DW(5); else DW(7);
#undef DW
  EY_LAUNCH_CHECK("ey_dwconv");
  return EY_OK;
}

extern "C" int ey_dwconv(int dtype, int B, int H, int W, int C, int k, int act, const void* x, int x_cstride, const void* w,
                         const float* bias, void* y, int y_cstride, ey_stream_t stream) {
  EY_CHECK(x && w && y, "dwconv: null pointer");
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "dwconv: bad dtype");
  EY_CHECK(k == 3 || k == 5 || k == 7, "dwconv: k=%d (3,5,7)", k);
  EY_CHECK(C > 0 && C % 8 == 0, "dwconv: C=%d must be a multiple of 8 (use ey_conv2d_direct)", C);
  const int es = dtype == EY_F16 ? 2 : 4;
  EY_CHECK(x_cstride >= C && y_cstride >= C && (x_cstride * es) % 16 == 0 && (y_cstride * es) % 16 == 0 && ey_aligned(x, 16) && ey_aligned(y, 16) && ey_aligned(w, 16),
           "dwconv: views must be 16-byte aligned");
  return dtype == EY_F16 ? dw_launch<f16>(B, H, W, C, k, act, x, x_cstride, w, bias, y, y_cstride, (hipStream_t)stream)
                         : dw_launch<float>(B, H, W, C, k, act, x, x_cstride, w, bias, y, y_cstride, (hipStream_t)stream);
}

// ============================================================================ Haar DWT (one level)
// taps = float32(1/sqrt2)^2 = 0.49999997 as the reference's pywt-derived conv weights (block.py:3597-3606)
template <typename T, int V>
__global__ __launch_bounds__(256) void dwt_kernel(int B, int H, int W, int C, const T* __restrict__ x, int xCs, T* __restrict__ y, int yCs) {
  const int Ho = H >> 1, Wo = W >> 1, cv = C / V;
  const long total = (long)B * Ho * Wo * cv;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c0 = (int)(idx % cv) * V;
  long m = idx / cv;
  const int ox = (int)(m % Wo);
  const long t = m / Wo;
  const int oy = (int)(t % Ho);
  const int b = (int)(t / Ho);
  const T* p00 = x + (((long)b * H + 2 * oy) * W + 2 * ox) * xCs + c0;
  const T* p01 = p00 + xCs;
  const T* p10 = p00 + (long)W * xCs;
  const T* p11 = p10 + xCs;
  T* yp = y + m * yCs + c0;
  const float s = 0.70710678118654752440f, tp = s * s;
  if constexpr (V == 8) {
    Vec8<T> a, bq, c, d, ll, lh, hl, hh;
    a.load(p00); bq.load(p01); c.load(p10); d.load(p11);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float fa = a.get(i) * tp, fb = bq.get(i) * tp, fc = c.get(i) * tp, fd = d.get(i) * tp;
      ll.set(i, (fa + fb) + (fc + fd));
      lh.set(i, (fa - fb) + (fc - fd));
      hl.set(i, (fa + fb) - (fc + fd));
      hh.set(i, (fa - fb) - (fc - fd));
    }
    ll.store(yp); lh.store(yp + C); hl.store(yp + 2 * C); hh.store(yp + 3 * C);
  } else {
    const float fa = to_f(*p00) * tp, fb = to_f(*p01) * tp, fc = to_f(*p10) * tp, fd = to_f(*p11) * tp;
    yp[0] = from_f<T>((fa + fb) + (fc + fd));
    yp[C] = from_f<T>((fa - fb) + (fc - fd));
    yp[2 * C] = from_f<T>((fa + fb) - (fc + fd));
    yp[3 * C] = from_f<T>((fa - fb) - (fc - fd));
  }
}

extern "C" int ey_dwt_haar(int dtype, int B, int H, int W, int C, const void* x, int x_cstride, void* y, int y_cstride, ey_stream_t stream) {
  EY_CHECK(x && y, "dwt: null pointer");
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "dwt: bad dtype");
  EY_CHECK(B > 0 && H >= 2 && W >= 2 && C > 0, "dwt: bad extent");
  EY_CHECK(x_cstride >= C && y_cstride >= 4 * C, "dwt: cstride");
  const int es = dtype == EY_F16 ? 2 : 4;
  const bool vec = C % 8 == 0 && (x_cstride * es) % 16 == 0 && (y_cstride * es) % 16 == 0 && ey_aligned(x, 16) && ey_aligned(y, 16);
  const long total = (long)B * (H / 2) * (W / 2) * (vec ? C / 8 : C);
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EY_F16) {
    if (vec) hipLaunchKernelGGL((dwt_kernel<f16, 8>), grid, dim3(256), 0, st, B, H, W, C, (const f16*)x, x_cstride, (f16*)y, y_cstride);
    else hipLaunchKernelGGL((dwt_kernel<f16, 1>), grid, dim3(256), 0, st, B, H, W, C, (const f16*)x, x_cstride, (f16*)y, y_cstride);
  } else {
    if (vec) hipLaunchKernelGGL((dwt_kernel<float, 8>), grid, dim3(256), 0, st, B, H, W, C, (const float*)x, x_cstride, (float*)y, y_cstride);
    else hipLaunchKernelGGL((dwt_kernel<float, 1>), grid, dim3(256), 0, st, B, H, W, C, (const float*)x, x_cstride, (float*)y, y_cstride);
  }
  EY_LAUNCH_CHECK("ey_dwt_haar");
  return EY_OK;
}

// ============================================================================ SPPF: three chained 5x5 max pools
// block = one image x CV consecutive 8-channel vectors; the whole (H x W x 8*CV ch) plane lives in LDS; separable max (row pass,
// column pass) applied three times, exactly the reference's chain (padding acts as -inf).  CV = 8 for f16 at 20x20 (64 channels =
// one 128-byte line per pixel: every line of x is fetched by exactly one workgroup and every output line is written whole; the first
// version, CV = 1, had 16 workgroups pull the same lines for their own 16-byte slices: 3.5x the algorithmic bytes at the L2's memory
// side, profiles/r02_pmc_traffic.json); smaller CV when the plane would not fit LDS.
template <typename T, int CV>
__global__ __launch_bounds__(1024) void sppf_kernel(int H, int W, int C, const T* __restrict__ x, int xCs, T* __restrict__ y1, T* __restrict__ y2,
                                                    T* __restrict__ y3, int yCs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Vec8<T>* A = reinterpret_cast<Vec8<T>*>(smem);
  Vec8<T>* Bf = A + H * W * CV;
  const int groups = C / (8 * CV);
  const int b = blockIdx.x / groups, c0 = (blockIdx.x % groups) * 8 * CV;
  const long base = (long)b * H * W;
  const int n = H * W * CV;  // element i = pixel i / CV, vector i % CV (consecutive threads: consecutive 16-byte pieces of a pixel's line)
  for (int i = threadIdx.x; i < n; i += blockDim.x) A[i].load(x + (base + i / CV) * xCs + c0 + (i % CV) * 8);
  __syncthreads();
  T* outs[3] = {y1, y2, y3};
  for (int pass = 0; pass < 3; ++pass) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {  // rows
      const int px = i / CV, yy = px / W, xx = px - yy * W;
      Vec8<T> mx = A[i];
      for (int dx = -2; dx <= 2; ++dx) {
        const int x2 = xx + dx;
        if (dx == 0 || x2 < 0 || x2 >= W) continue;
        const Vec8<T>& o = A[i + dx * CV];
#pragma unroll
        for (int k = 0; k < 8; ++k) mx.set(k, fmaxf(mx.get(k), o.get(k)));
      }
      Bf[i] = mx;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {  // columns
      const int px = i / CV, yy = px / W;
      Vec8<T> mx = Bf[i];
      for (int dy = -2; dy <= 2; ++dy) {
        const int y2 = yy + dy;
        if (dy == 0 || y2 < 0 || y2 >= H) continue;
        const Vec8<T>& o = Bf[i + dy * W * CV];
#pragma unroll
        for (int k = 0; k < 8; ++k) mx.set(k, fmaxf(mx.get(k), o.get(k)));
      }
      mx.store(outs[pass] + (base + px) * yCs + c0 + (i % CV) * 8);
      A[i] = mx;  // only element i of A is touched by this thread: no hazard with the column reads of Bf
    }
    __syncthreads();
  }
}

template <typename T, int CV>
static int sppf_launch(int B, int H, int W, int C, const void* x, int xCs, void* y1, void* y2, void* y3, int yCs, size_t lds, hipStream_t st) {
  static size_t reserved = 0;
  if (lds > 64 * 1024 && lds > reserved) {
    if (hipFuncSetAttribute((const void*)sppf_kernel<T, CV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "sppf: cannot reserve %zu B of LDS", lds);
    reserved = lds;
  }
  const int n = H * W * CV;
  const int threads = n >= 1024 ? 1024 : n >= 512 ? 512 : 256;
  hipLaunchKernelGGL((sppf_kernel<T, CV>), dim3(B * (C / (8 * CV))), dim3(threads), lds, st, H, W, C, (const T*)x, xCs, (T*)y1, (T*)y2, (T*)y3, yCs);
  return EY_OK;
}

extern "C" int ey_sppf_pool(int dtype, int B, int H, int W, int C, const void* x, int x_cstride, void* y1, void* y2, void* y3, int y_cstride,
                            ey_stream_t stream) {
  EY_CHECK(x && y1 && y2 && y3, "sppf: null pointer");
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "sppf: bad dtype");
  EY_CHECK(C > 0 && C % 8 == 0, "sppf: C=%d must be a multiple of 8", C);
  const int es = dtype == EY_F16 ? 2 : 4;
  EY_CHECK((size_t)H * W * 8 * es * 2 <= 160 * 1024, "sppf: %dx%d plane does not fit LDS", H, W);
  EY_CHECK((x_cstride * es) % 16 == 0 && (y_cstride * es) % 16 == 0 && ey_aligned(x, 16) && ey_aligned(y1, 16) && ey_aligned(y2, 16) && ey_aligned(y3, 16),
           "sppf: views must be 16-byte aligned");
  // widest channel group (whole 128-byte lines where possible) whose two planes fit LDS and that divides C
  int cv = 128 / (8 * es);  // vectors per 128-byte line: 8 (f16) / 4 (f32)
  while (cv > 1 && (C % (8 * cv) || (size_t)H * W * cv * 8 * es * 2 > 150 * 1024)) cv >>= 1;
  // ... but keep at least ~128 workgroups in flight: one CU stores its three output planes at a few bytes per clock, so 64 workgroups of
  // whole lines (23 us) lose to 128 of half lines (measured: profiles/r03_sppf_groups.txt)
  while (cv > 1 && (long)B * (C / (8 * cv)) < tune().sppf_min_wg) cv >>= 1;
  if (tune().sppf_cv > 0 && tune().sppf_cv <= cv) cv = (int)tune().sppf_cv;
  const size_t lds = (size_t)H * W * cv * 8 * es * 2;
  hipStream_t st = (hipStream_t)stream;
  int rc;
#define EY_SPPF(TT, CVV) rc = sppf_launch<TT, CVV>(B, H, W, C, x, x_cstride, y1, y2, y3, y_cstride, lds, st)
  if (dtype == EY_F16) { if (cv == 8) EY_SPPF(f16, 8); else if (cv == 4) EY_SPPF(f16, 4); else if (cv == 2) EY_SPPF(f16, 2); else EY_SPPF(f16, 1); }
  else { if (cv == 4) EY_SPPF(float, 4); else if (cv == 2) EY_SPPF(float, 2); else EY_SPPF(float, 1); }
#undef EY_SPPF
  if (rc != EY_OK) return rc;
  EY_LAUNCH_CHECK("ey_sppf_pool");
  return EY_OK;
}

// ============================================================================ slice copy (+ nearest x2) and layout transposes
template <typename T, int V>
__global__ __launch_bounds__(256) void copy_kernel(int B, int H, int W, int C, int up, const T* __restrict__ s, int sCs, T* __restrict__ d, int dCs) {
  const int cv = C / V;
  const long total = (long)B * H * W * cv;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c0 = (int)(idx % cv) * V;
  long m = idx / cv;
  const int xx = (int)(m % W);
  const long t = m / W;
  const int yy = (int)(t % H);
  const int b = (int)(t / H);
  const int Hs = H >> up, Ws = W >> up;
  const T* sp = s + (((long)b * Hs + (yy >> up)) * Ws + (xx >> up)) * sCs + c0;
  T* dp = d + m * dCs + c0;
  if constexpr (V == 8) { Vec8<T> v; v.load(sp); v.store(dp); }
  else *dp = *sp;
}

extern "C" int ey_copy_nhwc(int dtype, int B, int H, int W, int C, int up, const void* src, int src_cstride, void* dst, int dst_cstride, ey_stream_t stream) {
  EY_CHECK(src && dst, "copy: null pointer");
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "copy: bad dtype");
  EY_CHECK(up == 0 || (up == 1 && H % 2 == 0 && W % 2 == 0), "copy: up=%d needs even output extent", up);
  EY_CHECK(src_cstride >= C && dst_cstride >= C, "copy: cstride");
  const int es = dtype == EY_F16 ? 2 : 4;
  const bool vec = C % 8 == 0 && (src_cstride * es) % 16 == 0 && (dst_cstride * es) % 16 == 0 && ey_aligned(src, 16) && ey_aligned(dst, 16);
  const long total = (long)B * H * W * (vec ? C / 8 : C);
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EY_F16) {
    if (vec) hipLaunchKernelGGL((copy_kernel<f16, 8>), grid, dim3(256), 0, st, B, H, W, C, up, (const f16*)src, src_cstride, (f16*)dst, dst_cstride);
    else hipLaunchKernelGGL((copy_kernel<f16, 1>), grid, dim3(256), 0, st, B, H, W, C, up, (const f16*)src, src_cstride, (f16*)dst, dst_cstride);
  } else {
    if (vec) hipLaunchKernelGGL((copy_kernel<float, 8>), grid, dim3(256), 0, st, B, H, W, C, up, (const float*)src, src_cstride, (float*)dst, dst_cstride);
    else hipLaunchKernelGGL((copy_kernel<float, 1>), grid, dim3(256), 0, st, B, H, W, C, up, (const float*)src, src_cstride, (float*)dst, dst_cstride);
  }
  EY_LAUNCH_CHECK("ey_copy_nhwc");
  return EY_OK;
}

// tiled transpose through LDS: planes of (C x HW) <-> (HW x C)
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(int rows, int cols, const T* __restrict__ s, long sBatch, int sRow, T* __restrict__ d, long dBatch, int dRow) {
  // s[b][r][c] (row stride sRow) -> d[b][c][r] (row stride dRow)
  __shared__ T tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    if (r < rows && c < cols) tile[i][tx] = s[b * sBatch + (long)r * sRow + c];
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) d[b * dBatch + (long)c * dRow + r] = tile[tx][i];
  }
}

extern "C" int ey_nchw_to_nhwc(int dtype, int B, int C, int H, int W, const void* src, void* dst, int dst_cstride, ey_stream_t stream) {
  EY_CHECK(src && dst && dst_cstride >= C, "nchw_to_nhwc: bad args");
  const int HW = H * W;
  dim3 grid((HW + 31) / 32, (C + 31) / 32, B);
  if (dtype == EY_F16) hipLaunchKernelGGL(transpose_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, C, HW, (const f16*)src, (long)C * HW, HW, (f16*)dst, (long)HW * dst_cstride, dst_cstride);
  else if (dtype == EY_F32) hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, C, HW, (const float*)src, (long)C * HW, HW, (float*)dst, (long)HW * dst_cstride, dst_cstride);
  else return ey_set_error(EY_EINVAL, "nchw_to_nhwc: bad dtype");
  EY_LAUNCH_CHECK("ey_nchw_to_nhwc");
  return EY_OK;
}

extern "C" int ey_nhwc_to_nchw(int dtype, int B, int C, int H, int W, const void* src, int src_cstride, void* dst, ey_stream_t stream) {
  EY_CHECK(src && dst && src_cstride >= C, "nhwc_to_nchw: bad args");
  const int HW = H * W;
  dim3 grid((C + 31) / 32, (HW + 31) / 32, B);
  if (dtype == EY_F16) hipLaunchKernelGGL(transpose_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, HW, C, (const f16*)src, (long)HW * src_cstride, src_cstride, (f16*)dst, (long)C * HW, HW);
  else if (dtype == EY_F32) hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, HW, C, (const float*)src, (long)HW * src_cstride, src_cstride, (float*)dst, (long)C * HW, HW);
  else return ey_set_error(EY_EINVAL, "nhwc_to_nchw: bad dtype");
  EY_LAUNCH_CHECK("ey_nhwc_to_nchw");
  return EY_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// GPU pre-processing: LetterBox (data/augment.py:1556-1591: cv2.resize INTER_LINEAR + copyMakeBorder 114)
// + BasePredictor.preprocess (engine/predictor.py:123-133: BGR->RGB, HWC->CHW, uint8 -> f16/f32, /255) as ONE kernel from
// the raw uint8 image.  The resize restates OpenCV's 8-bit INTER_LINEAR fixed-point path (resize.cpp: 11-bit
// coefficients `saturate_cast<short>(w*2048)`, horizontal pass in int32, vertical pass
// `(((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2`) so the uint8 stage is integer-exact and identical to the CPU
// oracle; cv2 itself is not in the build image, so equality with cv2 is "parity unpinned" (see oracle/letterbox.py).
struct LbCoef { int s0, s1, a0, a1; };
__device__ __forceinline__ LbCoef lb_coef(int d, double scale, int ssize) {
  // fx = (float)((dx+0.5)*scale - 0.5); sx = floor(fx); fx -= sx  (no fma: the oracle evaluates the same IEEE operations)
  float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
  int s = (int)floorf(f);
  f = __fsub_rn(f, (float)s);
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  LbCoef c;
  c.s0 = s;
  c.s1 = min(s + 1, ssize - 1);
  c.a0 = __float2int_rn(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
  c.a1 = __float2int_rn(__fmul_rn(f, 2048.f));
  return c;
}

template <typename T>
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ src, int sh, int sw, int srow, long src_img, T* __restrict__ dst, int H, int W,
                                                        int nh, int nw, int top, int left, int pad, int swap_rb, double scale_x, double scale_y) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  src += (long)blockIdx.z * src_img;       // image blockIdx.z of a batch of same-shape images (ey_letterbox_batch)
  dst += (long)blockIdx.z * 3 * H * W;
  int v[3] = {pad, pad, pad};
  const int dx = x - left, dy = y - top;
  if (dx >= 0 && dx < nw && dy >= 0 && dy < nh) {
    if (nh == sh && nw == sw) {  // LetterBox skips cv2.resize when the size is unchanged
      const uint8_t* p = src + (long)dy * srow + dx * 3;
      v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    } else if (sw == 2 * nw && sh == 2 * nh) {  // cv::resize turns an exact 2x INTER_LINEAR decimation into the 2x2 box average
      const uint8_t* r0 = src + (long)(2 * dy) * srow + 6 * dx;
      const uint8_t* r1 = r0 + srow;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = (r0[c] + r0[3 + c] + r1[c] + r1[3 + c] + 2) >> 2;
    } else {
      const LbCoef cx = lb_coef(dx, scale_x, sw);
      // the vertical pass keeps its fractional weights at the borders and clamps the two rows (resizeGeneric_Invoker)
      float fy = (float)__dsub_rn(__dmul_rn((double)dy + 0.5, scale_y), 0.5);
      const int sy = (int)floorf(fy);
      fy = __fsub_rn(fy, (float)sy);
      const int b0 = __float2int_rn(__fmul_rn(__fsub_rn(1.f, fy), 2048.f)), b1 = __float2int_rn(__fmul_rn(fy, 2048.f));
      const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
      const uint8_t* r0 = src + (long)y0 * srow;
      const uint8_t* r1 = src + (long)y1 * srow;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int h0 = r0[cx.s0 * 3 + c] * cx.a0 + r0[cx.s1 * 3 + c] * cx.a1;
        const int h1 = r1[cx.s0 * 3 + c] * cx.a0 + r1[cx.s1 * 3 + c] * cx.a1;
        const int o = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        v[c] = min(max(o, 0), 255);
      }
    }
  }
  const long plane = (long)H * W, o = (long)y * W + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = swap_rb ? 2 - c : c;  // output plane c takes source channel sc (BGR -> RGB)
    dst[c * plane + o] = from_f<T>(__fdiv_rn((float)v[sc], 255.f));
  }
}

// Pure conversion (image already has the network's shape: no resize, no padding): 4 pixels per thread -- 12 source bytes as three
// aligned 32-bit loads, one 8/16-byte store per colour plane; grid-stride, so that a source in PINNED HOST memory (read over PCIe by
// the kernel itself) can be walked by a SMALL persistent grid: PCIe needs ~100 KB in flight, not 12 800 workgroups of waves that sit
// in the CUs' wave slots for microseconds per load and keep every other stream's kernels out (measured: the transfer then serialises
// with the whole pipeline exactly like a DMA copy does).
template <typename T>
__global__ __launch_bounds__(256) void u8hwc_to_chw_kernel(const uint8_t* __restrict__ src, int srow, long src_img, T* __restrict__ dst, int B, int H, int W, int swap_rb) {
  const int W4 = W >> 2;
  const long total = (long)B * H * W4, plane = (long)H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x4 = (int)(i % W4);
    const long t = i / W4;
    const int y = (int)(t % H), b = (int)(t / H);
    const unsigned* p = reinterpret_cast<const unsigned*>(src + (long)b * src_img + (long)y * srow + x4 * 12);
    const unsigned w0 = p[0], w1 = p[1], w2 = p[2];
    const unsigned char by[12] = {(unsigned char)w0, (unsigned char)(w0 >> 8), (unsigned char)(w0 >> 16), (unsigned char)(w0 >> 24),
                                  (unsigned char)w1, (unsigned char)(w1 >> 8), (unsigned char)(w1 >> 16), (unsigned char)(w1 >> 24),
                                  (unsigned char)w2, (unsigned char)(w2 >> 8), (unsigned char)(w2 >> 16), (unsigned char)(w2 >> 24)};
    T* d = dst + (long)b * 3 * plane + (long)y * W + x4 * 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int sc = swap_rb ? 2 - c : c;
      T o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = from_f<T>(__fdiv_rn((float)by[3 * q + sc], 255.f));
      if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(d + c * plane) = *reinterpret_cast<const uint2*>(o);
      else *reinterpret_cast<uint4*>(d + c * plane) = *reinterpret_cast<const uint4*>(o);
    }
  }
}

// true when p points into host memory (pinned, mapped): kernels that read it are PCIe-bound and get a small persistent grid
static bool ey_is_host_ptr(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

// Linear copy, 16 bytes per thread and iteration, grid-stride (the upload of a pinned host batch as an ordinary kernel).
__global__ __launch_bounds__(256) void linear_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int ey_copy_linear(const void* src, void* dst, size_t nbytes, ey_stream_t stream) {
  EY_CHECK(src && dst && nbytes % 16 == 0 && ey_aligned(src, 16) && ey_aligned(dst, 16), "copy_linear: 16-byte aligned pointers and size");
  if (!nbytes) return EY_OK;
  const long n16 = (long)(nbytes / 16);
  long g = (n16 + 255) / 256;
  const long cap = ey_is_host_ptr(src) ? 128 : 256L * 16;
  if (g > cap) g = cap;
  hipLaunchKernelGGL(linear_copy_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, n16);
  EY_LAUNCH_CHECK("ey_copy_linear");
  return EY_OK;
}

extern "C" int ey_letterbox_batch(int out_dtype, const uint8_t* src_hwc, int B, int src_h, int src_w, int src_row_bytes, long src_image_bytes, void* dst_chw, int H,
                                  int W, int new_h, int new_w, int top, int left, int pad_value, int swap_rb, ey_stream_t stream) {
  EY_CHECK(src_hwc && dst_chw, "letterbox: null pointer");
  EY_CHECK(B > 0 && B <= 65535 && src_image_bytes >= (long)src_h * src_row_bytes, "letterbox: batch of %d images, image pitch %ld bytes", B, src_image_bytes);
  EY_CHECK(out_dtype == EY_F16 || out_dtype == EY_F32, "letterbox: bad dtype");
  EY_CHECK(src_h > 0 && src_w > 0 && src_row_bytes >= 3 * src_w && H > 0 && W > 0, "letterbox: bad extent");
  EY_CHECK(new_h > 0 && new_w > 0 && top >= 0 && left >= 0 && top + new_h <= H && left + new_w <= W, "letterbox: resized image (%dx%d at %d,%d) outside the %dx%d canvas",
           new_h, new_w, top, left, H, W);
  EY_CHECK(pad_value >= 0 && pad_value <= 255, "letterbox: pad value");
  // cv::resize: inv_scale = dsize/ssize (double); scale = 1/inv_scale
  const double scale_x = 1.0 / ((double)new_w / (double)src_w), scale_y = 1.0 / ((double)new_h / (double)src_h);
  if (new_h == src_h && new_w == src_w && H == src_h && W == src_w && W % 4 == 0 && src_row_bytes % 4 == 0 && src_image_bytes % 4 == 0 && ey_aligned(src_hwc, 4) &&
      ey_aligned(dst_chw, 16)) {
    long g = ((long)B * H * (W / 4) + 255) / 256;
    const long cap = ey_is_host_ptr(src_hwc) ? 128 : 256L * 16;
    if (g > cap) g = cap;
    if (out_dtype == EY_F16)
      hipLaunchKernelGGL(u8hwc_to_chw_kernel<f16>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src_hwc, src_row_bytes, src_image_bytes, (f16*)dst_chw, B, H, W, swap_rb);
    else
      hipLaunchKernelGGL(u8hwc_to_chw_kernel<float>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src_hwc, src_row_bytes, src_image_bytes, (float*)dst_chw, B, H, W, swap_rb);
    EY_LAUNCH_CHECK("ey_letterbox(convert)");
    return EY_OK;
  }
  dim3 grid((W + 255) / 256, H, B);
  if (out_dtype == EY_F16)
    hipLaunchKernelGGL(letterbox_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, src_hwc, src_h, src_w, src_row_bytes, src_image_bytes, (f16*)dst_chw, H, W, new_h, new_w,
                       top, left, pad_value, swap_rb, scale_x, scale_y);
  else
    hipLaunchKernelGGL(letterbox_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src_hwc, src_h, src_w, src_row_bytes, src_image_bytes, (float*)dst_chw, H, W, new_h,
                       new_w, top, left, pad_value, swap_rb, scale_x, scale_y);
  EY_LAUNCH_CHECK("ey_letterbox");
  return EY_OK;
}

extern "C" int ey_letterbox(int out_dtype, const uint8_t* src_hwc, int src_h, int src_w, int src_row_bytes, void* dst_chw, int H, int W, int new_h,
                            int new_w, int top, int left, int pad_value, int swap_rb, ey_stream_t stream) {
  return ey_letterbox_batch(out_dtype, src_hwc, 1, src_h, src_w, src_row_bytes, (long)src_h * src_row_bytes, dst_chw, H, W, new_h, new_w, top, left, pad_value, swap_rb,
                            stream);
}
