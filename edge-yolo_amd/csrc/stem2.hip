// Layers 0 + 1 of the backbone as ONE kernel (f16):  y = Conv3x3s2_{16->C1}( Conv3x3s2_{3->16}(image) ), both + folded BN + SiLU
// (reference nn/modules/conv.py:41-59 twice; yolo11-test.yaml / EdgeLine yaml layers 0-1).
// The stem's (B,16,H/2,W/2) output is the largest tensor of the whole forward (105 MB at batch 32, 640x640) and has exactly one
// consumer; written and re-read it costs more HBM time than the image itself.  Here a 256-thread workgroup owns an 8 x TW tile of
// LAYER-1 output pixels:
//   A. the 3 x 35 x (4 TW + 8) input patch goes to LDS with aligned 16-byte loads of the planar image (range-checked: zero padding);
//   B. the 17 x (2 TW + 1) stem pixels the tile needs: per 16 pixels every lane gathers its 8 taps (k = c*9 + ky*3 + kx, OIHW order)
//      with 2-byte LDS reads, ONE 16x16x32 MFMA (K = 27 -> 32), bias + SiLU, rounded to f16 into an LDS tile -- zeros where the stem
//      pixel lies outside the stem map (= layer 1's padding).  Recomputed halo: 1 row / 1 column per tile (+ 9 %);
//   C. layer 1 from that tile: nine 16x16x16 MFMAs per 16 pixels per 16 channels (tap order, register-resident weights), the B
//      fragment of a tap = one 8-byte LDS read; bias + SiLU, 16-byte NHWC stores.
// Arithmetic = stem_mfma_kernel followed by conv3r_kernel, operation for operation: bit-identical to the two-launch form.
#include "common.h"
#include "tune.h"
#include <type_traits>

typedef f16 f16x4v __attribute__((ext_vector_type(4)));
#define ST2_SP 20  // pitch (halves) of a stem pixel's 16 channels in LDS: 40 B
template <int TH, int TW, int NT1>
__global__ __launch_bounds__(256) void stem_pair_kernel(int B, int H, int W, int Hs, int Ws, int Ho, int Wo, const f16* __restrict__ x, unsigned xbytes,
                                                        const float* __restrict__ w0, const float* __restrict__ bias0, const f16* __restrict__ w1, int Kpad1,
                                                        const float* __restrict__ bias1, f16* __restrict__ y, int yCs) {
  constexpr int IR = 4 * TH + 3, IV = (4 * TW + 8) / 8, LW = 4 * TW + 16, NV = 3 * IR * IV;  // input patch: rows, 16-byte vectors per row, LDS row pitch
  constexpr int SR = 2 * TH + 1, SW = 2 * TW + 1;                                            // stem tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* s_in = reinterpret_cast<f16*>(smem);  // [3][IR][LW]
  f16* s_st = s_in + 3 * IR * LW;            // [SR][SW][ST2_SP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int tilesX = (Wo + TW - 1) / TW, tilesY = (Ho + TH - 1) / TH;
  const int bid = (int)ey_xcd_block(blockIdx.x, gridDim.x);  // neighbouring tiles (shared patch rows / columns) in one XCD's L2
  const int b = bid / (tilesX * tilesY), trem = bid - b * (tilesX * tilesY);
  const int oy0 = (trem / tilesX) * TH, ox0 = (trem % tilesX) * TW;
  const int iy0 = 4 * oy0 - 3, ixa = 4 * ox0 - 8;  // first staged input row / (8-aligned) column
  // ---- A. input patch
  const __amdgpu_buffer_rsrc_t rs = ey_rsrc(x, xbytes);
#pragma unroll
  for (int u = 0; u < (NV + 255) / 256; ++u) {
    const int v = tid + u * 256;
    if (v < NV) {
      const int c = v / (IR * IV), rem = v - c * (IR * IV), row = rem / IV, vc = rem - row * IV;
      const int iy = iy0 + row, ix = ixa + 8 * vc;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;  // W % 8 == 0: a vector is entirely inside or entirely outside
      Vec8<f16> t;
      BufLoad8<f16>::load(t, rs, ok ? (unsigned)((((b * 3 + c) * H + iy) * W + ix) * 2) : EY_OOB);
      t.store(s_in + (c * IR + row) * LW + 8 * vc);
    }
  }
  // stem A fragment (fp32 OIHW -> f16) and this lane's tap offsets: k = 8g + t = c*9 + ky*3 + kx
  Vec8<f16> a0;
  int off[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int k = 8 * g + t, c = k / 9, ky = (k - 9 * c) / 3, kx = k - 9 * c - 3 * ky;
    off[t] = k < 27 ? (c * IR + ky) * LW + kx : 0;
    a0.v[t] = k < 27 ? (f16)w0[r * 27 + k] : (f16)0.f;
  }
  float bs0[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bs0[j] = bias0[4 * g + j];
  // layer-1 A fragments: lane (r, g) holds k = tap*16 + 4g .. +3 of packed weight row nt*16 + r
  f16x4v a1[9][NT1];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) a1[tap][nt] = *reinterpret_cast<const f16x4v*>(w1 + (nt * 16 + r) * Kpad1 + tap * 16 + 4 * g);
  __syncthreads();
  // ---- B. stem pixels of the tile, 16 per MFMA.  Blocks = 16-pixel row segments (SR rows x NSEG segments) + the last column as
  // ceil(SR / 16) column blocks; block id = wave + 4 it, so a wave's segments sit at a FIXED row stride: the 8 gather addresses of a lane
  // and its store address are bases + compile-time offsets (LDS immediates: no address arithmetic inside the unrolled loop).  Tiles whose
  // stem pixels all lie inside the stem map (all but the border tiles) skip the padding mask.
  constexpr int NSEG = TW / 8, RPI = 4 / NSEG, NRS = SR * NSEG, NFULL = NRS / 4, NB0 = NRS + (SR + 15) / 16, NIT0 = (NB0 + 3) / 4;
  static_assert(NSEG == 2 || NSEG == 4, "TW must be 16 or 32");
  const bool interior = 2 * oy0 - 1 >= 0 && 2 * oy0 - 1 + SR <= Hs && 2 * ox0 - 1 >= 0 && 2 * ox0 - 1 + SW <= Ws;  // (workgroup-uniform)
  auto stem_block = [&](const f16* const (&gp)[8], int goff, f16* dst, int row, int col, bool store, auto masked) {
    Vec8<f16> bq;
#pragma unroll
    for (int q = 0; q < 8; ++q) bq.v[q] = gp[q][goff];
    const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0.v, bq.v, (f32x4)0.f, 0, 0, 0);
    float v4[4] = {acc[0] + bs0[0], acc[1] + bs0[1], acc[2] + bs0[2], acc[3] + bs0[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) v4[j] = ey_silu_rn(v4[j]);
    f16x4v o = {(f16)v4[0], (f16)v4[1], (f16)v4[2], (f16)v4[3]};
    if constexpr (decltype(masked)::value) {  // stem pixels outside the stem map are layer 1's zero padding
      const int sy = 2 * oy0 - 1 + row, sx = 2 * ox0 - 1 + col;
      if (!(sy >= 0 && sy < Hs && sx >= 0 && sx < Ws)) o = (f16x4v)(f16)0.f;
    }
    if (store) *reinterpret_cast<f16x4v*>(dst) = o;
  };
  auto stem_phase = [&](auto masked) {
    {  // full iterations: every wave has a row segment
      const int row0 = wave / NSEG, col0 = (wave % NSEG) * 16 + r;
      const f16* gp[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) gp[q] = s_in + (2 * row0) * LW + 2 * col0 + 5 + off[q];
      f16* sp0 = s_st + (row0 * SW + col0) * ST2_SP + 4 * g;
#pragma unroll
      for (int it = 0; it < NFULL; ++it) stem_block(gp, it * (2 * RPI * LW), sp0 + it * (RPI * SW * ST2_SP), row0 + it * RPI, col0, true, masked);
    }
#pragma unroll
    for (int it = NFULL; it < NIT0; ++it) {  // the last row segments and the column blocks: per-lane coordinates
      const int id = wave + 4 * it;
      if (id >= NB0) break;  // (wave-uniform)
      int row, col;
      if (id < NRS) { row = id / NSEG; col = (id % NSEG) * 16 + r; }
      else { row = (id - NRS) * 16 + r; col = 2 * TW; }
      const bool valid = row < SR;
      row = valid ? row : SR - 1;
      const f16* gp[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) gp[q] = s_in + (2 * row) * LW + 2 * col + 5 + off[q];
      stem_block(gp, 0, s_st + (row * SW + col) * ST2_SP + 4 * g, row, col, valid, masked);
    }
  };
  if (interior) stem_phase(std::false_type{});
  else stem_phase(std::true_type{});
  __syncthreads();
  // ---- C. layer 1: 16-pixel row segments of the tile
  const int ch0 = g * 4 * NT1;
  float bs1[4 * NT1];
#pragma unroll
  for (int i = 0; i < 4 * NT1; ++i) bs1[i] = bias1[ch0 + i];
  constexpr int NBX = TW / 16;
  constexpr int NB1 = TH * NBX, NIT1 = (NB1 + 3) / 4;
#pragma unroll
  for (int it = 0; it < NIT1; ++it) {
    const int blk = wave + 4 * it;
    if (NB1 % 4 != 0 && blk >= NB1) break;
    const int ty = blk / NBX, tx = (blk - ty * NBX) * 16 + r;
    const f16* sp = s_st + ((2 * ty) * SW + 2 * tx) * ST2_SP + 4 * g;
    f16x4v bq[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) bq[ky * 3 + kx] = *reinterpret_cast<const f16x4v*>(sp + (ky * SW + kx) * ST2_SP);
    f32x4 acc[NT1];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) acc[nt] = (f32x4)0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1[tap][nt], bq[tap], acc[nt], 0, 0, 0);
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= Ho || ox >= Wo) continue;
    float v[4 * NT1];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * nt + j] = acc[nt][j] + bs1[4 * nt + j];
#pragma unroll
    for (int i = 0; i < 4 * NT1; ++i) v[i] = ey_silu_rn(v[i]);
    f16* yp = y + (((long)b * Ho + oy) * Wo + ox) * yCs + ch0;
#pragma unroll
    for (int h = 0; h < NT1 / 2; ++h) {
      Vec8<f16> o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.set(j, v[8 * h + j]);
      o.store(yp + 8 * h);
    }
  }
}

template <int TH, int TW, int NT1>
static int stem_pair_launch(int B, int H, int W, const void* x, const float* w0, const float* b0, const void* w1, int Kpad1, const float* b1, void* y,
                            int yCs, hipStream_t st) {
  const int Hs = (H - 1) / 2 + 1, Ws = (W - 1) / 2 + 1, Ho = (Hs - 1) / 2 + 1, Wo = (Ws - 1) / 2 + 1;
  const long tiles = (long)B * ((Wo + TW - 1) / TW) * ((Ho + TH - 1) / TH);
  const size_t lds = ((size_t)3 * (4 * TH + 3) * (4 * TW + 16) + (size_t)(2 * TH + 1) * (2 * TW + 1) * ST2_SP) * sizeof(f16);
  static bool reserved = false;
  if (lds > 64 * 1024 && !reserved) {
    if (hipFuncSetAttribute((const void*)stem_pair_kernel<TH, TW, NT1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "stem_pair: cannot reserve %zu B of LDS", lds);
    reserved = true;
  }
  hipLaunchKernelGGL((stem_pair_kernel<TH, TW, NT1>), dim3((unsigned)tiles), dim3(256), lds, st, B, H, W, Hs, Ws, Ho, Wo, (const f16*)x,
                     (unsigned)((long)B * 3 * H * W * 2), w0, b0, (const f16*)w1, Kpad1, b1, (f16*)y, yCs);
  EY_LAUNCH_CHECK("ey_stem_pair");
  return EY_OK;
}

extern "C" int ey_stem_pair(int B, int H, int W, const void* x_nchw, const float* w0_oihw, const float* bias0, int act0, int C1, const void* w1_packed,
                            const float* bias1, int act1, void* y, int y_cstride, ey_stream_t stream) {
  EY_CHECK(x_nchw && w0_oihw && w1_packed && y, "stem_pair: null pointer");
  EY_CHECK(B > 0 && H > 0 && W > 0, "stem_pair: bad extent");
  const int Hs = (H - 1) / 2 + 1, Ws = (W - 1) / 2 + 1, Ho = (Hs - 1) / 2 + 1, Wo = (Ws - 1) / 2 + 1;
  const bool fits = tune().stem_pair && C1 == 32 && act0 == EY_ACT_SILU && act1 == EY_ACT_SILU && bias0 && bias1 && W % 8 == 0 && ey_aligned(x_nchw, 16) && ey_aligned(w1_packed, 8) && y_cstride >= C1 && (y_cstride * 2) % 16 == 0 &&
                    ey_aligned(y, 16) && (long)B * 3 * H * W * 2 < (1L << 31) && (long)B * (Ho / 4 + 1) * (Wo / 16 + 1) < (1L << 31);
  if (!fits) return ey_set_error(EY_EUNSUPPORTED, "stem_pair: shape outside the fused kernel (f16 image, W %% 8 == 0, 3 -> 16 -> 32, bias + SiLU)");
  const int Kpad1 = ey_conv_kpad(9 * 16, 2);
  hipStream_t st = (hipStream_t)stream;
  switch (tune().stem_pair) {  // tile of layer-1 output pixels per workgroup (tools/stem_pair_bench.py: 8x32 87 us, 8x16 74, 4x16 94, 4x32 73; two launches 108)
    case 5: return stem_pair_launch<8, 32, 2>(B, H, W, x_nchw, w0_oihw, bias0, w1_packed, Kpad1, bias1, y, y_cstride, st);
    case 2: return stem_pair_launch<8, 16, 2>(B, H, W, x_nchw, w0_oihw, bias0, w1_packed, Kpad1, bias1, y, y_cstride, st);
    case 3: return stem_pair_launch<4, 16, 2>(B, H, W, x_nchw, w0_oihw, bias0, w1_packed, Kpad1, bias1, y, y_cstride, st);
    case 4: return stem_pair_launch<4, 32, 2>(B, H, W, x_nchw, w0_oihw, bias0, w1_packed, Kpad1, bias1, y, y_cstride, st);
    default: return stem_pair_launch<8, 16, 2>(B, H, W, x_nchw, w0_oihw, bias0, w1_packed, Kpad1, bias1, y, y_cstride, st);
  }
}
