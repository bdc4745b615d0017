// A block's closing 1x1 conv + the stride-2 3x3 conv that follows it, as ONE kernel (f16):
//     y = SiLU( Conv3x3s2_{64->64}( SiLU( Conv1x1_{C0+C1 -> 64}( cat(src0, src1) ) + b1 ) ) + b2 )
// = DSC3K2_Wavelet.cv2 (reference block.py:3783-3788; C2f.cv2 block.py:357-396) followed by the next backbone layer, a Conv(64, 64, 3, 2)
// (conv.py:41-59) -- layers 2 and 3 of the EdgeLine / YOLO11 n-scale backbones.  The 64-channel map between them (105 MB at batch 32,
// 160x160) has one consumer; written, then fetched 1.56x by the 3x3 windows, it is the second-largest HBM item of the forward.
// One persistent 512-thread workgroup per CU:
//   * the whole [64][576] 3x3 weight block stays in LDS (row pitch 592 halves = 2 (mod 4) 16-byte units: conflict-free fragment reads);
//   * per tile of 8 x 16 output pixels: phase 1 computes the 17 x 33 mid pixels the tile's windows touch -- B fragments by range-checked
//     buffer loads of the two sources (the next tile's are requested before phase 2 and fly through it), register-resident 1x1 weights,
//     bias + SiLU, rounded to f16 into an LDS tile (pixel pitch 144 B: the stride-2 fragment reads of phase 2 step 18 units = 2 (mod 4));
//     pixels outside the map are written as zeros (= the 3x3 conv's padding);  phase 2: one 16-pixel row segment per wave, 18 k-steps x 4
//     MFMAs with both operands from LDS; bias + SiLU, 16-byte NHWC stores.
// Both convs run the k-steps of their stand-alone kernels (conv_pwr_kernel, conv3s_kernel) in the same order on the same f16 values:
// the result is bit-identical to the two-launch form.
#include "common.h"
#include "tune.h"

struct Pw3P {
  int B, H, W, Ho, Wo;
  const void* src0; int cs0, C0; unsigned bytes0;
  const void* src1; int cs1, C1; unsigned bytes1;
  const void* w1; int Kpad1; const float* b1;
  const void* w2; int Kpad2; const float* b2;
  void* y; int yCs;
  int tilesX, tilesY, ntile, skew;
};

#define PW3_TH 8
#define PW3_TW 16
#define PW3_CP 72    // mid-tile pixel pitch (halves)
#define PW3_LW 592   // LDS pitch of a 3x3 weight row (halves)

__global__ __launch_bounds__(512) void pw3_kernel(Pw3P p) {
  typedef f16 T;
  constexpr int NT = 4, SR = 2 * PW3_TH + 1, SW = 2 * PW3_TW + 1, NS = SR * SW, NB = (NS + 15) / 16, NIT = (NB + 7) / 8, CP = PW3_CP, LW = PW3_LW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* s_w2 = reinterpret_cast<T*>(smem);  // [64][LW]
  T* s_mid = s_w2 + 64 * LW;             // [NS][CP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  for (int v = tid; v < 64 * 72; v += 512) {  // 3x3 weights: once per workgroup (K = 576 = 72 vectors per row)
    const int row = v / 72, kv = v - row * 72;
    Vec8<T> w;
    w.load((const T*)p.w2 + (long)row * p.Kpad2 + kv * 8);
    w.store(s_w2 + row * LW + kv * 8);
  }
  // 1x1 weights: k-step 0 = source 0 (channels 8g.., valid while < C0), k-step 1 = source 1; packed row k = [src0 | src1]
  Vec8<T> af1[2][NT];
  {
    const __amdgpu_buffer_rsrc_t rw = ey_rsrc(p.w1, (unsigned)(64 * p.Kpad1 * 2));
    const unsigned wvoff = (unsigned)((r * p.Kpad1 + 8 * g) * 2);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) BufLoad8<T>::load(af1[t][nt], rw, wvoff, (nt * 16 * p.Kpad1 + (t ? p.C0 : 0)) * 2);
  }
  const int ch0 = g * 4 * NT;
  float bias1[4 * NT], bias2[4 * NT];
#pragma unroll
  for (int i = 0; i < 4 * NT; ++i) {
    bias1[i] = p.b1[ch0 + i];
    bias2[i] = p.b2[ch0 + i];
  }
  const bool cok0 = 8 * g < p.C0, cok1 = 8 * g < p.C1;
  const __amdgpu_buffer_rsrc_t rs0 = ey_rsrc(p.src0, p.bytes0), rs1 = ey_rsrc(p.src1, p.bytes1);
  const int tiles_img = p.tilesX * p.tilesY;

  // phase-1 operands of a tile: block it of this wave = mid pixels (wave + 8 it) * 16 + r of the flattened 17 x 33 tile
  Vec8<T> bf[NIT][2];
  auto issue = [&](int tile, unsigned& inmask) {
    const int b = tile / tiles_img, trem = tile - b * tiles_img;
    const int oy0 = (trem / p.tilesX) * PW3_TH, ox0 = (trem % p.tilesX) * PW3_TW;
    inmask = 0u;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pi = (wave + 8 * it) * 16 + r, pc = pi < NS ? pi : NS - 1;
      const int t = pc / SW, u = pc - t * SW;
      const int my = 2 * oy0 - 1 + t, mx = 2 * ox0 - 1 + u;
      const bool in = my >= 0 && my < p.H && mx >= 0 && mx < p.W;
      inmask |= in ? (1u << it) : 0u;
      const int m = (b * p.H + my) * p.W + mx;
      BufLoad8<T>::load(bf[it][0], rs0, (in && cok0) ? (unsigned)((m * p.cs0 + 8 * g) * 2) : EY_OOB);
      BufLoad8<T>::load(bf[it][1], rs1, (in && cok1) ? (unsigned)((m * p.cs1 + 8 * g) * 2) : EY_OOB);
    }
  };
  unsigned inmask = 0u;
  int tile = blockIdx.x;
  if (tile < p.ntile) issue(tile, inmask);
  __syncthreads();  // weights staged
  for (; tile < p.ntile; tile += gridDim.x) {
    const int b = tile / tiles_img, trem = tile - b * tiles_img;
    const int oy0 = (trem / p.tilesX) * PW3_TH, ox0 = (trem % p.tilesX) * PW3_TW;
    // ---- phase 1: the 1x1 conv on the tile's mid pixels -> LDS
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int blk = wave + 8 * it;
      if (NB % 8 != 0 && it == NIT - 1 && blk >= NB) break;  // (wave-uniform)
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4)0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1[t][nt].v, bf[it][t].v, acc[nt], 0, 0, 0);
      const int pi = blk * 16 + r;
      const bool in = (inmask >> it) & 1u;
      float v[4 * NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * nt + j] = ey_silu_rn(acc[nt][j] + bias1[4 * nt + j]);
      if (pi < NS) {
        T* d = s_mid + pi * CP + ch0;
        const unsigned keep = in ? 0xffffffffu : 0u;  // pixels outside the map are the 3x3 conv's zero padding (masking the packed words: 4 ops per 8 values)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          Vec8<T> o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o.set(j, v[8 * h + j]);
          ey_u32x4 w = __builtin_bit_cast(ey_u32x4, o.v);
          w &= keep;
          o.v = __builtin_bit_cast(f16x8, w);
          o.store(d + 8 * h);
        }
      }
    }
    ey_lds_barrier();
    unsigned inmask_next = 0u;
    if (tile + (int)gridDim.x < p.ntile) issue(tile + gridDim.x, inmask_next);  // in flight through phase 2
    // ---- phase 2: 3x3 stride 2 from the LDS tile; wave = output row `wave` of the tile, lane r = column
    {
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4)0.f;
      const T* bp = s_mid + ((2 * wave) * SW + 2 * r) * CP + 8 * g;
      const T* ap = s_w2 + r * LW + 8 * g;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          Vec8<T> bq, af[NT];
          bq.load(bp + (ky * SW + kx) * CP + ks * 32);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) af[nt].load(ap + nt * 16 * LW + tap * 64 + ks * 32);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[nt].v, bq.v, acc[nt], 0, 0, 0);
        }
      }
      const int oy = oy0 + wave, ox = ox0 + r;
      if (oy < p.Ho && ox < p.Wo) {
        float v[4 * NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[4 * nt + j] = ey_silu_rn(acc[nt][j] + bias2[4 * nt + j]);
        T* yp = (T*)p.y + (((long)b * p.Ho + oy) * p.Wo + ox) * p.yCs + ch0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          Vec8<T> o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o.set(j, v[8 * h + j]);
          o.store(yp + 8 * h);
        }
      }
    }
    inmask = inmask_next;
    ey_lds_barrier();  // phase 2 reads of the tile are done before the next phase 1 overwrites it
  }
}

// ---- second form: two independent 256-thread workgroups per CU, no weights in LDS.
// Wave w owns output channels 16w .. 16w + 15 of phase 2: its 18 A fragments (3x3 weights of those channels) live in registers, the B
// fragments come from the LDS mid tile (81 KB per workgroup: two workgroups fit a CU, so phase 1 of one -- VALU: 36 k SiLUs per tile --
// can run beside phase 2 of the other -- MFMA + LDS reads).  Phase 1 as above (a wave computes all 64 mid channels of its 9 pixel
// blocks).  Requesting the next tile's operands before phase 2 (72 more live registers) measured slower: 50.9 -> 55.5 us.
__global__ __launch_bounds__(256, 2) void pw3b_kernel(Pw3P p) {
  typedef f16 T;
  constexpr int NT = 4, SR = 2 * PW3_TH + 1, SW = 2 * PW3_TW + 1, NS = SR * SW, NB = (NS + 15) / 16, NIT = (NB + 3) / 4, CP = PW3_CP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* s_mid = reinterpret_cast<T*>(smem);  // [NS][CP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  // 3x3 weights of this wave's 16 channels: MFMA row r <-> channel 16 wave + r <-> packed row (r / 4) * 16 + 4 wave + r % 4  (NTpack = 4)
  Vec8<T> af2[18];
  {
    const T* wr = (const T*)p.w2 + (long)((r >> 2) * 16 + 4 * wave + (r & 3)) * p.Kpad2 + 8 * g;
#pragma unroll
    for (int s = 0; s < 18; ++s) af2[s].load(wr + s * 32);
  }
  Vec8<T> af1[2][NT];
  {
    const __amdgpu_buffer_rsrc_t rw = ey_rsrc(p.w1, (unsigned)(64 * p.Kpad1 * 2));
    const unsigned wvoff = (unsigned)((r * p.Kpad1 + 8 * g) * 2);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) BufLoad8<T>::load(af1[t][nt], rw, wvoff, (nt * 16 * p.Kpad1 + (t ? p.C0 : 0)) * 2);
  }
  const int ch0 = g * 4 * NT;
  float bias2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias2[j] = p.b2[16 * wave + 4 * g + j];
  const f32x4* b1v = reinterpret_cast<const f32x4*>(p.b1 + ch0);
  const bool cok0 = 8 * g < p.C0, cok1 = 8 * g < p.C1;
  const __amdgpu_buffer_rsrc_t rs0 = ey_rsrc(p.src0, p.bytes0), rs1 = ey_rsrc(p.src1, p.bytes1);
  const int tiles_img = p.tilesX * p.tilesY;
  // the two workgroups of a CU start together and every tile costs the same: left alone they run phase 1 (VALU) at the same time and
  // phase 2 (MFMA + LDS) at the same time.  The second half of the grid starts a fraction of a tile late.
  if (blockIdx.x >= gridDim.x / 2)
    for (int i = 0; i < p.skew; ++i) __builtin_amdgcn_s_sleep(16);
  // (a contiguous run of tiles per workgroup with XCD-contiguous runs measured slower here: 49.6 -> 55.4 us)
  for (int tile = blockIdx.x; tile < p.ntile; tile += gridDim.x) {
    const int b = tile / tiles_img, trem = tile - b * tiles_img;
    const int oy0 = (trem / p.tilesX) * PW3_TH, ox0 = (trem % p.tilesX) * PW3_TW;
    // ---- phase 1
    {
      Vec8<T> bf[NIT][2];
      unsigned inmask = 0u;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int pi = (wave + 4 * it) * 16 + r, pc = pi < NS ? pi : NS - 1;
        const int t = pc / SW, u = pc - t * SW;
        const int my = 2 * oy0 - 1 + t, mx = 2 * ox0 - 1 + u;
        const bool in = my >= 0 && my < p.H && mx >= 0 && mx < p.W;
        inmask |= in ? (1u << it) : 0u;
        const int m = (b * p.H + my) * p.W + mx;
        BufLoad8<T>::load(bf[it][0], rs0, (in && cok0) ? (unsigned)((m * p.cs0 + 8 * g) * 2) : EY_OOB);
        BufLoad8<T>::load(bf[it][1], rs1, (in && cok1) ? (unsigned)((m * p.cs1 + 8 * g) * 2) : EY_OOB);
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int blk = wave + 4 * it;
        if (NB % 4 != 0 && it == NIT - 1 && blk >= NB) break;  // (wave-uniform)
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4)0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1[t][nt].v, bf[it][t].v, acc[nt], 0, 0, 0);
        const int pi = blk * 16 + r;
        float v[4 * NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const f32x4 bb = b1v[nt];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[4 * nt + j] = ey_silu_rn(acc[nt][j] + bb[j]);
        }
        if (pi < NS) {
          T* d = s_mid + pi * CP + ch0;
          const unsigned keep = ((inmask >> it) & 1u) ? 0xffffffffu : 0u;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            Vec8<T> o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.set(j, v[8 * h + j]);
            ey_u32x4 w = __builtin_bit_cast(ey_u32x4, o.v);
            w &= keep;
            o.v = __builtin_bit_cast(f16x8, w);
            o.store(d + 8 * h);
          }
        }
      }
    }
    ey_lds_barrier();
    // ---- phase 2: all 8 row segments of the tile for this wave's 16 channels
#pragma unroll 2
    for (int mb = 0; mb < PW3_TH; ++mb) {
      f32x4 acc = (f32x4)0.f;
      const T* bp = s_mid + ((2 * mb) * SW + 2 * r) * CP + 8 * g;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          Vec8<T> bq;
          bq.load(bp + (ky * SW + kx) * CP + ks * 32);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af2[tap * 2 + ks].v, bq.v, acc, 0, 0, 0);
        }
      }
      const int oy = oy0 + mb, ox = ox0 + r;
      if (oy < p.Ho && ox < p.Wo) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ey_silu_rn(acc[j] + bias2[j]);
        const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        *reinterpret_cast<f16x4*>((T*)p.y + (((long)b * p.Ho + oy) * p.Wo + ox) * p.yCs + 16 * wave + 4 * g) = o;
      }
    }
    ey_lds_barrier();
  }
}

extern "C" int ey_conv_pw_conv3s2(int dtype, int B, int H, int W, const void* src0, int C0, int cstride0, const void* src1, int C1, int cstride1, int Cmid,
                                  const void* w1_packed, const float* bias1, int act1, int Cout, const void* w2_packed, const float* bias2, int act2, void* y,
                                  int y_cstride, ey_stream_t stream) {
  EY_CHECK(src0 && src1 && w1_packed && w2_packed && y, "conv_pw_conv3s2: null pointer");
  EY_CHECK(B > 0 && H > 0 && W > 0 && C0 > 0 && C1 > 0, "conv_pw_conv3s2: bad extent");
  EY_CHECK(cstride0 >= C0 && cstride1 >= C1 && y_cstride >= Cout, "conv_pw_conv3s2: cstride");
  const long px = (long)B * H * W;
  const long bytes0 = ((px - 1) * cstride0 + C0) * 2, bytes1 = ((px - 1) * cstride1 + C1) * 2;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const bool fits = tune().pw3 && dtype == EY_F16 && Cmid == 64 && Cout == 64 && C0 <= 32 && C1 <= 32 && C0 % 8 == 0 && C1 % 8 == 0 && act1 == EY_ACT_SILU &&
                    act2 == EY_ACT_SILU && bias1 && bias2 && px >= tune().pw3_min_px && bytes0 < (1L << 31) && bytes1 < (1L << 31) &&
                    (cstride0 * 2) % 16 == 0 && (cstride1 * 2) % 16 == 0 && ey_aligned(src0, 16) && ey_aligned(src1, 16) && (y_cstride * 2) % 16 == 0 && ey_aligned(y, 16) &&
                    ey_aligned(w1_packed, 16) && ey_aligned(w2_packed, 16);
  if (!fits) return ey_set_error(EY_EUNSUPPORTED, "conv_pw_conv3s2: shape outside the fused kernel (f16, <= 32 + <= 32 -> 64 -> 64, bias + SiLU, large maps)");
  Pw3P p;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
  p.src0 = src0; p.cs0 = cstride0; p.C0 = C0; p.bytes0 = (unsigned)bytes0;
  p.src1 = src1; p.cs1 = cstride1; p.C1 = C1; p.bytes1 = (unsigned)bytes1;
  p.w1 = w1_packed; p.Kpad1 = ey_conv_kpad(C0 + C1, 2); p.b1 = bias1;
  p.w2 = w2_packed; p.Kpad2 = ey_conv_kpad(9 * 64, 2); p.b2 = bias2;
  p.y = y; p.yCs = y_cstride;
  p.tilesX = (Wo + PW3_TW - 1) / PW3_TW; p.tilesY = (Ho + PW3_TH - 1) / PW3_TH;
  const long ntile = (long)B * p.tilesX * p.tilesY;
  if (ntile >= (1L << 30)) return ey_set_error(EY_EUNSUPPORTED, "conv_pw_conv3s2: too many tiles");
  p.ntile = (int)ntile;
  p.skew = (int)tune().pw3_skew;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu < 1) ncu = 256;
  }
  const size_t mid = (size_t)(2 * PW3_TH + 1) * (2 * PW3_TW + 1) * PW3_CP * 2;
  if (tune().pw3 == 2) {  // first form: one 512-thread workgroup per CU, 3x3 weights in LDS (measured 57.9 us; the other form 50.9)
    const size_t lds = (size_t)64 * PW3_LW * 2 + mid;
    static bool reserved = false;
    if (!reserved) {
      if (hipFuncSetAttribute((const void*)pw3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ey_set_error(EY_ELAUNCH, "conv_pw_conv3s2: cannot reserve %zu B of LDS", lds);
      reserved = true;
    }
    const int grid = ntile < ncu ? (int)ntile : ncu;
    hipLaunchKernelGGL(pw3_kernel, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, p);
  } else {
    static bool reserved = false;
    if (!reserved) {
      if (hipFuncSetAttribute((const void*)pw3b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid) != hipSuccess)
        return ey_set_error(EY_ELAUNCH, "conv_pw_conv3s2: cannot reserve %zu B of LDS", mid);
      reserved = true;
    }
    const int grid = ntile < 2 * ncu ? (int)ntile : 2 * ncu;
    hipLaunchKernelGGL(pw3b_kernel, dim3((unsigned)grid), dim3(256), mid, (hipStream_t)stream, p);
  }
  EY_LAUNCH_CHECK("ey_conv_pw_conv3s2");
  return EY_OK;
}
