// K4+K5 fused: the half-resolution branch of _WaveletEnhancer (reference nn/modules/block.py:3645-3710) in ONE kernel (f16):
//
//     b (B,H,W,c) --Haar DWT--> LL|LH|HL|HH (c each, H/2 x W/2)          _PywtDWT2D.forward, block.py:3619-3642
//       --f_ll 1x1 on LL, shared f_h 3x3 on LH/HL/HH (+BN folded, SiLU)--> P (2c)   block.py:3688-3694
//       --Z = W_z . P  (fuse's columns over the four processed sub-bands, band weights folded in)--> Z (B,H/2,W/2,c)
//
// Unfused this is three launches (dwt_kernel, the 4-group conv3_tile launch, a 1x1 conv) that write and re-read the 4c-channel
// sub-band tensor and the 2c-channel P at half resolution (PMC: 1.3-1.85x their algorithmic bytes through halo re-reads).  Here a
// 256-thread workgroup owns a TH x 16 tile of half-resolution pixels: it reads the (2TH+4) x 36 input patch once, keeps the four
// sub-bands (with the 3x3 halo, zero outside the map = the conv's padding) and P in LDS, and writes only Z.  Same rounding points
// as the unfused f16 path (sub-bands and P are f16 tensors there too), fp32 accumulation, the K order of ey_conv2d.
#include "common.h"
#include "tune.h"

struct WzP {
  int B, H, W, Ho, Wo;
  const f16* x; int xCs;
  const f16* w_sub; long w_set;   // two packed 3x3 weight sets [c/2][9c]: set 0 = f_ll as a centre-tap 3x3, set 1 = f_h
  const float* b_sub;             // [2][c/2]
  const f16* w_z;                 // packed 1x1 [c][2c]
  f16* z; int zCs;
  int tiles_x, tiles_y;
  int xcd;  // XCD-contiguous tile order (ey_xcd_block)
};

__host__ __device__ constexpr int wz_nt(int cout) { return cout <= 16 ? 1 : cout <= 32 ? 2 : cout <= 64 ? 4 : cout <= 80 ? 5 : 8; }
__host__ __device__ constexpr int wz_kpad(int K) {  // == ey_conv_kpad(K, 2): 2 (mod 4) 16-byte units
  int units = (K + 32) >> 3;
  while ((units & 3) != 2) ++units;
  return units * 8;
}

// NS: the output-channel blocks of both contractions are split over NS wave groups (4 x NS waves per workgroup): wide channel counts sit
// on small maps (c = 128 at 20x20: 3 tiles per image), where four waves walking 500 dependent MFMA steps each would be pure latency.
template <int C, int TH, int NS, bool GS>
__global__ __launch_bounds__(256 * NS) void wavelet_z_kernel(WzP p) {
  constexpr int TW = 16, RH = TH + 2, RW = TW + 2, NPOS = RH * RW, SS = C + 8, PS = 2 * C + 8, H2 = C / 2, CV = C / 8, NTHR = 256 * NS;
  constexpr int NTs_all = wz_nt(H2), NTz_all = wz_nt(C), MT = TH / 4;
  constexpr int NTs = NTs_all / NS, NTz = NTz_all / NS;
  static_assert(NTs >= 1 && NTs * NS == NTs_all && NTz * NS == NTz_all, "channel blocks must split evenly over the wave groups");
  constexpr int KPs = wz_kpad(9 * C), KPz = wz_kpad(2 * C);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* S = reinterpret_cast<f16*>(smem);   // [4][NPOS][SS]
  f16* P = S + 4 * NPOS * SS;               // [TH*16][PS]
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, ns = tid >> 8, r = lane & 15, g = lane >> 4;
  int blk = p.xcd ? (int)ey_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x;  // neighbouring tiles (shared halo patches) in one XCD's L2
  const int tx = blk % p.tiles_x;
  blk /= p.tiles_x;
  const int ty = blk % p.tiles_y, b = blk / p.tiles_y;
  const int ty0 = ty * TH, tx0 = tx * TW;
  // ---- 1. Haar sub-bands of the halo'd tile -> LDS (zero outside the half-resolution map: the 3x3 conv's padding).  All of a thread's
  // 2x2 input patches are requested before the first butterfly (one memory round trip instead of one per position).
  {
    const float sq = 0.70710678118654752440f, tp = sq * sq;
    constexpr int NIT = (NPOS * CV + NTHR - 1) / NTHR;
    const long xbytes = ((long)p.B * p.H * p.W - 1) * p.xCs * 2 + C * 2;
    const __amdgpu_buffer_rsrc_t rx = ey_rsrc(p.x, (unsigned)xbytes);  // (< 2 GiB: host check)
    Vec8<f16> in[NIT][4];
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
      const int it = tid + n * NTHR;
      const int pos = it / CV, c8 = (it - pos * CV) * 8;
      const int ry = pos / RW, rx_ = pos - ry * RW;
      const int hy = ty0 - 1 + ry, hx = tx0 - 1 + rx_;
      const bool ok = it < NPOS * CV && hy >= 0 && hy < p.Ho && hx >= 0 && hx < p.Wo;
      const unsigned o00 = ok ? (unsigned)(((((long)b * p.H + 2 * hy) * p.W + 2 * hx) * p.xCs + c8) * 2) : EY_OOB;  // out of range -> zeros
      BufLoad8<f16>::load(in[n][0], rx, o00);
      BufLoad8<f16>::load(in[n][1], rx, ok ? o00 + (unsigned)(p.xCs * 2) : EY_OOB);
      BufLoad8<f16>::load(in[n][2], rx, ok ? o00 + (unsigned)(p.W * p.xCs * 2) : EY_OOB);
      BufLoad8<f16>::load(in[n][3], rx, ok ? o00 + (unsigned)((p.W + 1) * p.xCs * 2) : EY_OOB);
    }
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
      const int it = tid + n * NTHR;
      if (it < NPOS * CV) {
        const int pos = it / CV, c8 = (it - pos * CV) * 8;
        Vec8<f16> ll, lh, hl, hh;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float fa = in[n][0].get(j) * tp, fb = in[n][1].get(j) * tp, fc = in[n][2].get(j) * tp, fd = in[n][3].get(j) * tp;
          ll.set(j, (fa + fb) + (fc + fd));
          lh.set(j, (fa - fb) + (fc - fd));
          hl.set(j, (fa + fb) - (fc + fd));
          hh.set(j, (fa - fb) - (fc - fd));
        }
        f16* sp = S + pos * SS + c8;
        ll.store(sp); lh.store(sp + NPOS * SS); hl.store(sp + 2 * NPOS * SS); hh.store(sp + 3 * NPOS * SS);
      }
    }
  }
  // ---- 2. the four sub-band convs -> P (f16, [pixel][LL | LH | HL | HH processed, c/2 each]).
  // GS = false: a wave owns MT pixel rows and the NTs channel blocks of its wave group, and walks the four sub-bands in turn;
  // GS = true (c = 128: 8 waves, 4-row tiles): a wave owns ONE sub-band, the NTs channel blocks of its wave group and all TH rows -- every weight fragment
  //   is then fetched by one wave instead of four (the per-wave fragment loads were 1.8 MB per workgroup through the CU's 64 B/clk
  //   vector-memory path) and feeds TH MFMAs instead of one.
  // Either way the wave's k-steps (LL: the centre tap's C/32 steps of the centre-tap 3x3 packing; LH/HL/HH: 9C/32 steps each) form ONE
  // sequence whose weight fragments are requested UB steps at a time, the next batch while the current one is multiplied: one exposed
  // L2 round trip per wave instead of one per batch (a lone batch is a dependent ~1.5 us trip; there were 7-10 per wave).
  {
    constexpr int MT2 = GS ? TH : MT, NT2 = NTs;
    constexpr int KS_ALL = (9 * C + 31) / 32, J0 = (4 * C) / 32, J1 = (5 * C + 31) / 32, N0 = J1 - J0;
    constexpr int UB = (NT2 * 9 <= 12) ? 9 : 12 / NT2;  // steps per batch: two batches of fragments (4 VGPRs each) are live
    const int wave16 = tid >> 6;
    const int mygrp = GS ? (wave16 & 3) : 0, nt0 = ns * NTs, row0 = GS ? 0 : wave * MT;
    const int T = GS ? (mygrp == 0 ? N0 : KS_ALL) : N0 + 3 * KS_ALL;
    const __amdgpu_buffer_rsrc_t rws = ey_rsrc(p.w_sub, (unsigned)((p.w_set + (long)16 * NTs_all * KPs) * 2));
    const unsigned wvoff = (unsigned)((r * KPs + 8 * g) * 2);
    auto decode = [&](int s_, int& grp, int& j) {  // (wave-uniform)
      if constexpr (GS) { grp = mygrp; j = (mygrp == 0 ? J0 : 0) + s_; }
      else if (s_ < N0) { grp = 0; j = J0 + s_; }
      else { const int t = s_ - N0; grp = 1 + t / KS_ALL; j = t - (grp - 1) * KS_ALL; }
    };
    auto issue = [&](int s0, Vec8<f16> (&af)[UB][NT2]) {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        int grp, j;
        decode(s0 + u, grp, j);
        const bool ok = s0 + u < T;
        const long setoff = grp == 0 ? 0 : p.w_set;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
          BufLoad8<f16>::load(af[u][nt], rws, ok ? wvoff + (unsigned)(setoff * 2) : EY_OOB, ok ? ((nt0 + nt) * 16 * KPs + 32 * j) * 2 : 0);
      }
    };
    f32x4 acc[MT2][NT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) acc[mt][nt] = (f32x4)0.f;
    auto compute = [&](int s0, const Vec8<f16> (&af)[UB][NT2]) {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (s0 + u < T) {  // (wave-uniform)
          int grp, j;
          decode(s0 + u, grp, j);
          const f16* Sg = S + grp * NPOS * SS;
          const int k0 = 32 * j + 8 * g, tap = k0 / C, ch = k0 - tap * C;  // (C is a power of two >= 16: 8 | C, a fragment never straddles taps)
          const int dy = tap / 3, dx = tap - dy * 3;
          Vec8<f16> bf[MT2];
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt) {
            if (tap < 9) bf[mt].load(Sg + ((row0 + mt + dy) * RW + r + dx) * SS + ch);
            else bf[mt].zero();
          }
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u][nt].v, bf[mt].v, acc[mt][nt], 0, 0, 0);
          if (j == (grp == 0 ? J1 : KS_ALL) - 1) {  // last k-step of this sub-band: bias + SiLU -> P, start the next sub-band from zero
            const float* bias = p.b_sub + (grp == 0 ? 0 : H2);
#pragma unroll
            for (int mt = 0; mt < MT2; ++mt) {
              f16* pp = P + ((row0 + mt) * 16 + r) * PS + grp * H2;
#pragma unroll
              for (int nt = 0; nt < NT2; ++nt) {
                const int ch0 = g * 4 * NTs_all + 4 * (nt0 + nt);
                if (ch0 < H2) {
                  float v[4];
#pragma unroll
                  for (int q = 0; q < 4; ++q) {
                    const float t = acc[mt][nt][q] + bias[ch0 + q];
                    v[q] = t * ey_sigmoid(t);  // SiLU (Conv.default_act)
                  }
                  const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                  *reinterpret_cast<f16x4*>(pp + ch0) = o;
                }
                acc[mt][nt] = (f32x4)0.f;
              }
            }
          }
        }
      }
    };
    Vec8<f16> afA[UB][NT2], afB[UB][NT2];
    issue(0, afA);    // (weights do not depend on the sub-bands: in flight across the barrier that publishes them)
    __syncthreads();
#pragma unroll 1
    for (int s0 = 0; s0 < T; s0 += 2 * UB) {
      issue(s0 + UB, afB);
      compute(s0, afA);
      issue(s0 + 2 * UB, afA);
      compute(s0 + UB, afB);
    }
  }
  // ---- 3. Z = W_z . P for this wave's pixels -> global (no bias, no activation: the pre-activation term of fuse, added there).
  // The first batch of W_z fragments does not depend on P: it is requested before the barrier that publishes P.
  {
    const __amdgpu_buffer_rsrc_t rwz = ey_rsrc(p.w_z, (unsigned)((long)16 * NTz_all * KPz * 2));
    const unsigned wzoff = (unsigned)((r * KPz + 8 * g) * 2);
    constexpr int KZ = (2 * C) / 32, UZ = (KZ * NTz <= 32) ? KZ : 32 / NTz;  // (<= 32 fragments = 128 VGPRs; the sub-band phase's registers are dead here)
    Vec8<f16> af[UZ][NTz];
    auto issue_z = [&](int jb) {
#pragma unroll
      for (int u = 0; u < UZ; ++u)
#pragma unroll
        for (int nt = 0; nt < NTz; ++nt) BufLoad8<f16>::load(af[u][nt], rwz, jb + u < KZ ? wzoff : EY_OOB, jb + u < KZ ? ((ns * NTz + nt) * 16 * KPz + 32 * (jb + u)) * 2 : 0);
    };
    issue_z(0);
    __syncthreads();
    f32x4 acc[MT][NTz];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTz; ++nt) acc[mt][nt] = (f32x4)0.f;
    for (int jb = 0; jb < KZ; jb += UZ) {
      if (jb) issue_z(jb);
#pragma unroll
      for (int u = 0; u < UZ; ++u) {
        if (jb + u < KZ) {
          Vec8<f16> bf[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) bf[mt].load(P + ((wave * MT + mt) * 16 + r) * PS + 32 * (jb + u) + 8 * g);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTz; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u][nt].v, bf[mt].v, acc[mt][nt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int hy = ty0 + wave * MT + mt, hx = tx0 + r;
      if (hy < p.Ho && hx < p.Wo) {
        f16* zp = p.z + (((long)b * p.Ho + hy) * p.Wo + hx) * p.zCs + g * 4 * NTz_all + 4 * ns * NTz;
#pragma unroll
        for (int nt = 0; nt < NTz; ++nt) {
          const f16x4 o = {(f16)acc[mt][nt][0], (f16)acc[mt][nt][1], (f16)acc[mt][nt][2], (f16)acc[mt][nt][3]};
          *reinterpret_cast<f16x4*>(zp + 4 * nt) = o;
        }
      }
    }
  }
}

template <int C, int TH, int NS, bool GS>
static int wz_launch(WzP p, hipStream_t st) {
  constexpr int NPOS = (TH + 2) * 18;
  const size_t lds = (size_t)(4 * NPOS * (C + 8) + TH * 16 * (2 * C + 8)) * 2;
  static bool attr_done = false;
  if (lds > 64 * 1024 && !attr_done) {
    if (hipFuncSetAttribute((const void*)wavelet_z_kernel<C, TH, NS, GS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "wavelet_z: cannot reserve %zu B of LDS", lds);
    attr_done = true;
  }
  p.tiles_x = (p.Wo + 15) / 16;
  p.tiles_y = (p.Ho + TH - 1) / TH;
  const long nblk = (long)p.B * p.tiles_x * p.tiles_y;
  if (nblk >= (1L << 31)) return ey_set_error(EY_EINVAL, "wavelet_z: too many tiles");
  p.xcd = (int)((tune().xcd_map >> 2) & 1);
  hipLaunchKernelGGL((wavelet_z_kernel<C, TH, NS, GS>), dim3((unsigned)nblk), dim3(256 * NS), lds, st, p);
  EY_LAUNCH_CHECK("ey_wavelet_z");
  return EY_OK;
}

extern "C" int ey_wavelet_z(int dtype, int B, int H, int W, int C, const void* x, int x_cstride, const void* w_sub_packed, long w_set_stride, const float* b_sub,
                            const void* w_z_packed, void* z, int z_cstride, ey_stream_t stream) {
  if (dtype != EY_F16) return ey_set_error(EY_EUNSUPPORTED, "wavelet_z: f16 only (the fp32 parity mode keeps the per-layer kernels)");
  if (!(C == 16 || C == 32 || C == 64 || C == 128)) return ey_set_error(EY_EUNSUPPORTED, "wavelet_z: built for c in {16, 32, 64, 128}, got %d", C);
  EY_CHECK(x && w_sub_packed && b_sub && w_z_packed && z && B > 0 && H >= 2 && W >= 2, "wavelet_z: bad arguments");
  EY_CHECK(x_cstride >= C && (x_cstride * 2) % 16 == 0 && ey_aligned(x, 16) && z_cstride >= C && (z_cstride * 2) % 8 == 0 && ey_aligned(z, 8),
           "wavelet_z: view alignment");
  EY_CHECK(ey_aligned(w_sub_packed, 16) && ey_aligned(w_z_packed, 16) && (w_set_stride * 2) % 16 == 0 && w_set_stride >= 0, "wavelet_z: weight alignment");
  EY_CHECK(((long)B * H * W - 1) * x_cstride * 2 + C * 2 < (1L << 31), "wavelet_z: input view of 2 GiB or more");
  WzP p;
  p.B = B; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
  p.x = (const f16*)x; p.xCs = x_cstride; p.w_sub = (const f16*)w_sub_packed; p.w_set = w_set_stride; p.b_sub = b_sub; p.w_z = (const f16*)w_z_packed;
  p.z = (f16*)z; p.zCs = z_cstride;
  hipStream_t st = (hipStream_t)stream;
  switch (C) {
    case 16: return wz_launch<16, 8, 1, false>(p, st);
    case 32: return wz_launch<32, 8, 1, false>(p, st);
    case 64: return wz_launch<64, 8, 2, false>(p, st);
    default: return wz_launch<128, 4, 2, true>(p, st);
  }
}
