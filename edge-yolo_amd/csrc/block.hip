// Block programs: a whole chain of layers on SMALL feature maps (20x20 at 640x640) as ONE launch.
//
// At batch 32 the 20x20 part of the network (stride-2 conv -> DSC3K2_Wavelet -> SPPF -> C2PSA_LinearAttention, the last neck
// block, the 20x20 head towers) is ~45 launches of 5-20 us that each move 1-6 MB: pure launch / ramp-up / tail latency.  Here ONE
// persistent 1024-thread workgroup per image walks a "program" of stages (conv 1x1/3x3 with the full ey_conv2d epilogue, depthwise
// kxk, Haar DWT, the SPPF pool chain, MFMA linear attention) with a workgroup barrier between stages.  Activations go through
// global memory (they stay in L2: 20x20x256 f16 = 200 KB per image), weights stream from L2, nothing returns to the host between
// layers; 32 images occupy 32 CUs and leave the other 224 to the pipeline stages that run beside it (large-map, HBM-bound layers).
//
// Same arithmetic as the stand-alone kernels: f16 storage between stages (the reference's intermediate tensors), fp32
// accumulation (v_mfma_f32_16x16x32_f16), fp32 epilogues; the K order of every contraction is (ky, kx, source, channel) like
// ey_conv2d.  f16 only (the fp32 parity mode keeps the per-layer kernels).
#include "common.h"
#include "linattn_mfma.inc.h"

#define BLK_THREADS 1024
#define BLK_WAVES 16
#define BLK_LDS_BYTES (104 * 1024)
#define BLK_MAX_EXT 8

typedef ey_block_stage BlkStage;
struct BlkExt { char* p[BLK_MAX_EXT]; };

__device__ __forceinline__ void load4(const f16* p, float (&o)[4]) {
  const f16x4 v = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (float)v[j];
}
__device__ __forceinline__ void store4(f16* p, const float* v) {
  const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  *reinterpret_cast<f16x4*>(p) = o;
}
__device__ __forceinline__ const f16* blk_ptr(const BlkExt& e, int64_t addr, int ext) {
  return reinterpret_cast<const f16*>(ext >= 0 ? e.p[ext] + addr : reinterpret_cast<char*>(addr));
}

// ---------------------------------------------------------------------------------------------------------------- conv epilogue
// ey_conv2d's: y = res + out_scale * act(acc + bias + bilinear(addz)) for one pixel (m = oy*Wo + ox) and the 4*NTI consecutive
// channels a lane owns; the pointers are already offset to the lane's first channel.  One 4-channel quad at a time (small live set).
struct BlkEpi {
  const float* bias;
  f16* y;
  const f16* res;
  const f16* z;
  int y_cs, res_cs, addz_cs, Hz, Wz, act;
  float zsy, zsx, out_scale;
};
template <int NTI>
__device__ __forceinline__ void blk_epilogue(const BlkEpi& e, const f32x4 (&acc)[NTI], int m, int oy, int ox, int yrow = -1, int rrow = -1) {
  // (yrow / rrow: row index inside an LDS-resident tile instead of the pixel index m, block_tile_kernel)
  long z00 = 0, z01 = 0, z10 = 0, z11 = 0;
  float ly0 = 0.f, ly1 = 0.f, lx0 = 0.f, lx1 = 0.f;
  if (e.z) {  // F.interpolate(size=(Ho,Wo), bilinear, align_corners=False): ATen area_pixel_compute_source_index, scale = in/out
    const float sy = fmaxf(e.zsy * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(e.zsx * (ox + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, e.Hz - 1), x1 = min(x0 + 1, e.Wz - 1);
    ly1 = sy - y0; lx1 = sx - x0; ly0 = 1.f - ly1; lx0 = 1.f - lx1;
    z00 = (long)(y0 * e.Wz + x0) * e.addz_cs; z01 = (long)(y0 * e.Wz + x1) * e.addz_cs;
    z10 = (long)(y1 * e.Wz + x0) * e.addz_cs; z11 = (long)(y1 * e.Wz + x1) * e.addz_cs;
  }
  f16* yp = e.y + (long)(yrow >= 0 ? yrow : m) * e.y_cs;
  const f16* rp = e.res ? e.res + (long)(rrow >= 0 ? rrow : m) * e.res_cs : nullptr;
#pragma unroll
  for (int q = 0; q < NTI; ++q) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = acc[q][j];
    if (e.bias) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(e.bias + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += bv[j];
    }
    if (e.z) {
      float a00[4], a01[4], a10[4], a11[4];
      load4(e.z + z00 + 4 * q, a00); load4(e.z + z01 + 4 * q, a01); load4(e.z + z10 + 4 * q, a10); load4(e.z + z11 + 4 * q, a11);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += ly0 * (lx0 * a00[j] + lx1 * a01[j]) + ly1 * (lx0 * a10[j] + lx1 * a11[j]);
    }
    if (e.act == EY_ACT_SILU && e.out_scale == 1.f) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = v[j] * ey_sigmoid(v[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = ey_act(v[j], e.act) * e.out_scale;
    }
    if (rp) {
      float rr[4];
      load4(rp + 4 * q, rr);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += rr[j];
    }
    store4(yp + 4 * q, v);
  }
}

// ---------------------------------------------------------------------------------------------------------------- conv
// Work item = (group, MT pixel tiles of 16, NTI of the NT 16-channel MFMA row blocks of one packed block tile); the 16 waves walk
// the items round-robin, consecutive waves share a pixel tile (its fragments hit in L1).  A = packed weights (row permutation of
// ey_conv_pack_weight: lane (r, g) ends up with 4*NTI CONSECUTIVE channels of pixel r), B = pixels by range-checked buffer loads
// (padding taps, pixel tails and channel tails read zeros).
template <int MT, int NTI>
__device__ void blk_conv(const BlkStage& sg, const BlkExt& ext, int b) {
  // the stage descriptor lives in global memory: every field the loops use is read ONCE into (scalar) registers here
  const int H = sg.H, W = sg.W, Wo = sg.Wo, kk = sg.k, stride = sg.stride, nsrc = sg.nsrc, Cout = sg.Cout, kpad = sg.kpad;
  const int C0 = sg.src_C[0], C1 = sg.src_C[1], cs0 = sg.src_cs[0], cs1 = sg.src_cs[1];
  const int ngroup = sg.ngroup, w_gmax = sg.w_gmax, act = sg.act;
  const long src_g = sg.src_g, y_g = sg.y_g, w_g = sg.w_g;
  const f16* src0 = blk_ptr(ext, sg.src[0], sg.src_ext[0]) + (long)b * sg.src_img[0];
  const f16* src1 = nsrc > 1 ? blk_ptr(ext, sg.src[1], sg.src_ext[1]) + (long)b * sg.src_img[1] : src0;
  const f16* wptr = reinterpret_cast<const f16*>(sg.w);
  const float* bptr = reinterpret_cast<const float*>(sg.bias);
  f16* yptr = const_cast<f16*>(blk_ptr(ext, sg.y, sg.y_ext)) + (long)b * sg.y_img;
  const int y_cs = sg.y_cs, res_cs = sg.res_cs, addz_cs = sg.addz_cs, Hz = sg.addz_H, Wz = sg.addz_W;
  const f16* resptr = sg.has_res ? blk_ptr(ext, sg.res, sg.res_ext) + (long)b * sg.res_img : nullptr;
  const f16* zptr = sg.has_addz ? blk_ptr(ext, sg.addz, sg.addz_ext) + (long)b * sg.addz_img : nullptr;
  const float zsy = sg.zsy, zsx = sg.zsx, out_scale = sg.out_scale;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int M = sg.Ho * Wo, NT = sg.nt_pack, BN = 16 * NT;
  const int mtiles = (M + 16 * MT - 1) / (16 * MT);
  const int nsub = NT / NTI, nblk = (Cout + BN - 1) / BN;
  const int nitems = ngroup * mtiles * nblk * nsub;
  const int pad = kk >> 1;
  const unsigned bytes0 = (unsigned)((((long)H * W - 1) * cs0 + C0) * 2), bytes1 = (unsigned)((((long)H * W - 1) * cs1 + C1) * 2);
  for (int it = wave; it < nitems; it += BLK_WAVES) {
    const int sub = it % nsub;
    int t = it / nsub;
    const int nb = t % nblk;
    t /= nblk;
    const int mt_i = t % mtiles, grp = t / mtiles;
    const int wset = min(grp, w_gmax);
    // ---- operands
    __amdgpu_buffer_rsrc_t rs[2];
    rs[0] = ey_rsrc(src0 + (long)grp * src_g, bytes0);
    rs[1] = ey_rsrc(src1, bytes1);
    const __amdgpu_buffer_rsrc_t rw = ey_rsrc(wptr + (long)wset * w_g, (unsigned)((long)nblk * BN * kpad * 2));
    unsigned woff[NTI];
#pragma unroll
    for (int nt = 0; nt < NTI; ++nt) woff[nt] = (unsigned)((((nb * BN + (sub * NTI + nt) * 16 + r) * kpad) + 8 * g) * 2);
    int oy[MT], ox[MT];
    bool pv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = (mt_i * MT + mt) * 16 + r;
      pv[mt] = m < M;
      const int mm = pv[mt] ? m : 0;
      oy[mt] = mm / Wo;
      ox[mt] = mm - oy[mt] * Wo;
    }
    f32x4 acc[MT][NTI];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTI; ++nt) acc[mt][nt] = (f32x4)0.f;
    // ---- K loop: k-steps = (ky, kx, source, 32-channel step), walked U at a time: the loads of U steps are issued back to back
    // (a lone step is one dependent round trip to L2, ~1 us: the loop would be pure latency), then their MFMAs
    constexpr int U = (MT + NTI <= 3) ? 4 : (MT + NTI <= 5) ? 3 : 2;
    int ky = 0, kx = 0, si = 0, c0 = 0, kofs = 0;
    bool more = true;
    while (more) {
      Vec8<f16> bf[U][MT], af[U][NTI];
      bool live[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        live[u] = more;
        if (more) {  // (wave-uniform)
          const int C = si == 0 ? C0 : C1, cs2 = (si == 0 ? cs0 : cs1) * 2;
          const bool chan_ok = c0 + 8 * g < C;  // channel tail of a source that is not a multiple of 32: zeros, not the neighbour's data
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int iy = oy[mt] * stride - pad + ky, ix = ox[mt] * stride - pad + kx;
            const bool inb = chan_ok && pv[mt] && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const unsigned po = inb ? (unsigned)(iy * W + ix) * (unsigned)cs2 + 16u * g : EY_OOB;
            if (si == 0) BufLoad8<f16>::load(bf[u][mt], rs[0], po, c0 * 2);
            else BufLoad8<f16>::load(bf[u][mt], rs[1], po, c0 * 2);
          }
#pragma unroll
          for (int nt = 0; nt < NTI; ++nt) BufLoad8<f16>::load(af[u][nt], rw, woff[nt], (kofs + c0) * 2);
          // advance the (wave-uniform) k-step iterator
          c0 += 32;
          if (c0 >= C) {
            kofs += C;
            c0 = 0;
            if (++si >= nsrc) {
              si = 0;
              if (++kx >= kk) {
                kx = 0;
                if (++ky >= kk) more = false;
              }
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (live[u]) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTI; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u][nt].v, bf[u][mt].v, acc[mt][nt], 0, 0, 0);
        }
      }
    }
    // ---- epilogue
    const int ch0 = nb * BN + g * 4 * NT + 4 * sub * NTI;
    if (ch0 >= Cout) continue;
    BlkEpi e;
    e.bias = bptr ? bptr + wset * Cout + ch0 : nullptr;
    e.y = yptr + (long)grp * y_g + ch0;
    e.res = resptr ? resptr + (long)grp * y_g + ch0 : nullptr;
    e.z = zptr ? zptr + (long)grp * y_g + ch0 : nullptr;
    e.y_cs = y_cs; e.res_cs = res_cs; e.addz_cs = addz_cs; e.Hz = Hz; e.Wz = Wz; e.act = act; e.zsy = zsy; e.zsx = zsx; e.out_scale = out_scale;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (pv[mt]) blk_epilogue<NTI>(e, acc[mt], (mt_i * MT + mt) * 16 + r, oy[mt], ox[mt]);
  }
}

// ---------------------------------------------------------------------------------------------------------------- depthwise
// y[p][c] = act(sum_taps x[p + tap][c] * w[tap][c] + bias[c]), k x k, stride 1, pad k/2; thread item = (pixel, 8 channels).
// DSConv.dw (conv.py:94-97,102; no bias, no activation, result rounded to f16 like the reference's intermediate tensor) and DWConv.
__device__ void blk_dw(const BlkStage& s, const BlkExt& ext, int b) {
  const int C = s.src_C[0], cv = C >> 3, H = s.H, W = s.W, HW = H * W, k = s.k, pad = k >> 1, xcs = s.src_cs[0], ycs = s.y_cs, act = s.act;
  const f16* xb = blk_ptr(ext, s.src[0], s.src_ext[0]) + (long)b * s.src_img[0];
  const __amdgpu_buffer_rsrc_t rx = ey_rsrc(xb, (unsigned)((((long)HW - 1) * xcs + C) * 2));
  const f16* w = reinterpret_cast<const f16*>(s.w);
  const float* bias = reinterpret_cast<const float*>(s.bias);
  f16* yb = const_cast<f16*>(blk_ptr(ext, s.y, s.y_ext)) + (long)b * s.y_img;
  for (int i = threadIdx.x; i < HW * cv; i += BLK_THREADS) {
    const int c8 = (i % cv) * 8, p = i / cv;
    const int py = p / W, px = p - py * W;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int ky = 0; ky < k; ++ky) {
      const int iy = py - pad + ky;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = px - pad + kx;
        const bool inb = iy >= 0 && iy < H && ix >= 0 && ix < W;
        Vec8<f16> xv, wv;
        BufLoad8<f16>::load(xv, rx, inb ? (unsigned)(((iy * W + ix) * xcs + c8) * 2) : EY_OOB);
        wv.load(w + (ky * k + kx) * C + c8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += xv.get(j) * wv.get(j);
      }
    }
    Vec8<f16> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.set(j, ey_act(acc[j] + (bias ? bias[c8 + j] : 0.f), act));
    o.store(yb + (long)p * ycs + c8);
  }
}

// ---------------------------------------------------------------------------------------------------------------- Haar DWT
// _PywtDWT2D.forward (block.py:3619-3642), same arithmetic as dwt_kernel: taps float32(1/sqrt2)^2; y = [LL | LH | HL | HH] channel blocks.
__device__ void blk_dwt(const BlkStage& s, const BlkExt& ext, int b) {
  const int C = s.src_C[0], cv = C >> 3, Ho = s.Ho, Wo = s.Wo, W = s.W, xcs = s.src_cs[0], ycs = s.y_cs;
  const f16* xb = blk_ptr(ext, s.src[0], s.src_ext[0]) + (long)b * s.src_img[0];
  f16* yb = const_cast<f16*>(blk_ptr(ext, s.y, s.y_ext)) + (long)b * s.y_img;
  const float sq = 0.70710678118654752440f, tp = sq * sq;
  for (int i = threadIdx.x; i < Ho * Wo * cv; i += BLK_THREADS) {
    const int c8 = (i % cv) * 8, m = i / cv;
    const int oy = m / Wo, ox = m - oy * Wo;
    const f16* p00 = xb + (long)((2 * oy) * W + 2 * ox) * xcs + c8;
    Vec8<f16> a, bq, c, d, ll, lh, hl, hh;
    a.load(p00); bq.load(p00 + xcs); c.load(p00 + (long)W * xcs); d.load(p00 + (long)(W + 1) * xcs);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float fa = a.get(j) * tp, fb = bq.get(j) * tp, fc = c.get(j) * tp, fd = d.get(j) * tp;
      ll.set(j, (fa + fb) + (fc + fd));
      lh.set(j, (fa - fb) + (fc - fd));
      hl.set(j, (fa + fb) - (fc + fd));
      hh.set(j, (fa - fb) - (fc - fd));
    }
    f16* yp = yb + (long)m * ycs + c8;
    ll.store(yp); lh.store(yp + C); hl.store(yp + 2 * C); hh.store(yp + 3 * C);
  }
}

// ---------------------------------------------------------------------------------------------------------------- SPPF pool chain
// y1 = mp5(x), y2 = mp5(y1), y3 = mp5(y2) (block.py:219-223; padding acts as -inf): per chunk of channel octets the H x W planes
// live in LDS (two buffers), separable max (row pass, column pass) three times; y_k = y + k*C channels of the concat buffer.
__device__ void blk_pool(const BlkStage& s, const BlkExt& ext, int b, char* smem) {
  const int C = s.src_C[0], cv = C >> 3, H = s.H, W = s.W, HW = H * W, xcs = s.src_cs[0], ycs = s.y_cs;
  const f16* xb = blk_ptr(ext, s.src[0], s.src_ext[0]) + (long)b * s.src_img[0];
  f16* yb = const_cast<f16*>(blk_ptr(ext, s.y, s.y_ext)) + (long)b * s.y_img;
  const int cvc = max(1, min(cv, (BLK_LDS_BYTES / 2) / (HW * 16)));  // channel octets per chunk (host checked that one fits)
  Vec8<f16>* A = reinterpret_cast<Vec8<f16>*>(smem);
  Vec8<f16>* Bf = A + cvc * HW;
  for (int cg0 = 0; cg0 < cv; cg0 += cvc) {
    const int ng = min(cvc, cv - cg0);
    __syncthreads();  // the previous chunk's LDS reads are done
    for (int i = threadIdx.x; i < ng * HW; i += BLK_THREADS) {
      const int cg = i % ng, p = i / ng;
      A[cg * HW + p].load(xb + (long)p * xcs + (cg0 + cg) * 8);
    }
    __syncthreads();
    for (int pass = 0; pass < 3; ++pass) {
      for (int i = threadIdx.x; i < ng * HW; i += BLK_THREADS) {  // rows
        const int cg = i / HW, p = i - cg * HW, yy = p / W, xx = p - yy * W;
        const Vec8<f16>* row = A + cg * HW + yy * W;
        Vec8<f16> mx = row[xx];
        for (int dx = -2; dx <= 2; ++dx) {
          const int x2 = xx + dx;
          if (dx == 0 || x2 < 0 || x2 >= W) continue;
          const Vec8<f16> o = row[x2];
#pragma unroll
          for (int j = 0; j < 8; ++j) mx.set(j, fmaxf(mx.get(j), o.get(j)));
        }
        Bf[i] = mx;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < ng * HW; i += BLK_THREADS) {  // columns
        const int cg = i / HW, p = i - cg * HW, yy = p / W, xx = p - yy * W;
        const Vec8<f16>* pl = Bf + cg * HW;
        Vec8<f16> mx = pl[p];
        for (int dy = -2; dy <= 2; ++dy) {
          const int y2 = yy + dy;
          if (dy == 0 || y2 < 0 || y2 >= H) continue;
          const Vec8<f16> o = pl[y2 * W + xx];
#pragma unroll
          for (int j = 0; j < 8; ++j) mx.set(j, fmaxf(mx.get(j), o.get(j)));
        }
        mx.store(yb + (long)p * ycs + (long)pass * C + (cg0 + cg) * 8);  // y = the y1 slot; y2, y3 follow it at C-channel steps
        A[i] = mx;  // only element i of A is touched by this thread
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- linear attention
// LinearAttention.forward core (block.py:3360-3373) for head_dim 64: two heads side by side (512 threads each).
__device__ void blk_linattn(const BlkStage& s, const BlkExt& ext, int b, char* smem) {
  LinAttnLds* S = reinterpret_cast<LinAttnLds*>(smem);
  const int N = s.H * s.W, C = s.Cout, half = threadIdx.x >> 9, tid = threadIdx.x & 511;
  const f16* qkv = blk_ptr(ext, s.src[0], s.src_ext[0]) + (long)b * s.src_img[0];
  f16* y = const_cast<f16*>(blk_ptr(ext, s.y, s.y_ext)) + (long)b * s.y_img;
  for (int h0 = 0; h0 < s.heads; h0 += 2) {
    const int h = h0 + half;
    __syncthreads();  // LDS of the previous pair is free
    const f16* qb = qkv + h * 64;
    linattn_mfma_head(S[half], N, qb, qb + C, qb + 2 * C, s.src_cs[0], y + h * 64, s.y_cs, tid);
  }
}

// ---------------------------------------------------------------------------------------------------------------- the kernel
template <int MT>
__device__ __forceinline__ void blk_conv_nti(const BlkStage& s, const BlkExt& ext, int b) {
  switch (s.nti) {
    case 1: blk_conv<MT, 1>(s, ext, b); break;
    case 2: blk_conv<MT, 2>(s, ext, b); break;
    case 4: if constexpr (MT <= 2) blk_conv<MT, 4>(s, ext, b); break;
    case 5: if constexpr (MT == 1) blk_conv<MT, 5>(s, ext, b); break;
    default: break;
  }
}

__global__ __launch_bounds__(BLK_THREADS) void block_kernel(const BlkStage* __restrict__ prog, int nstages, BlkExt ext, long long* __restrict__ tstamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  if (tstamps && b == 0 && threadIdx.x == 0) tstamps[0] = wall_clock64();
  for (int si = 0; si < nstages; ++si) {
    const BlkStage& s = prog[si];
    switch (s.op) {
      case EY_BLK_CONV:
        if (s.mt == 1) blk_conv_nti<1>(s, ext, b);
        else if (s.mt == 2) blk_conv_nti<2>(s, ext, b);
        else blk_conv_nti<4>(s, ext, b);
        break;
      case EY_BLK_DW: blk_dw(s, ext, b); break;
      case EY_BLK_DWT: blk_dwt(s, ext, b); break;
      case EY_BLK_POOL: blk_pool(s, ext, b, smem); break;
      case EY_BLK_LINATTN: blk_linattn(s, ext, b, smem); break;
      default: break;
    }
    __syncthreads();  // this stage's global writes are visible to the whole workgroup (one CU, one L1) before the next stage reads them
    if (tstamps && b == 0 && threadIdx.x == 0) tstamps[si + 1] = wall_clock64();  // (developer timing, ey_block_run_timed)
  }
}

// ---------------------------------------------------------------------------------------------------------------- pointwise chains
// A chain of 1x1 convs (every stage: k = 1, stride 1, one group, the same map) has no coupling between pixels, so it needs no
// per-image workgroup: one 256-thread workgroup per TILE_PX = 32 pixels walks the stages; its 4 waves split the output-channel blocks
// of a stage (2 pixel tiles x NTI row blocks per item: every weight fragment feeds two MFMAs).  Built for a short dependent chain per
// stage: the stage descriptors are resolved ONCE into LDS, intermediates that only the chain reads live in LDS (rows of C + 8
// halves: conflict-free 16-byte reads), the weight fragments of a whole item are requested in batches of 8 k-steps, and only the
// chain's outputs go to global memory.  C2PSA's proj -> ffn -> ffn -> cv2 tail, cv1 -> qkv, the enhancer tail -> DSC3k.cv1|cv2,
// DSC3k.cv3 -> DSC3K2.cv2 (block.py:3412-3497,1506-1562,3749-3788).
#define TILE_PX 32
#define TILE_MAX_STAGES 8
struct TStage {
  const f16* src[2]; const f16* w; const float* bias; f16* y; const f16* res; const f16* z;
  int src_lds[2], src_cs[2], src_C[2], nsrc;     // src_lds >= 0: element offset of the source's pixel row 0 in LDS (row stride src_cs)
  int y_lds, y_cs, res_lds, res_cs;               // same for the output / residual (y_lds >= 0: the stage writes LDS only)
  int kpad, nt_pack, nti, Cout, act, addz_cs, Hz, Wz;
  float zsy, zsx, out_scale;
};

template <int NTI>
__device__ __forceinline__ void tile_conv(const TStage& t, f16* lds, int b, int m0, int M, int Wo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int NT = t.nt_pack, BN = 16 * NT, nsub = NT / NTI, nblk = (t.Cout + BN - 1) / BN, nitems = nblk * nsub;
  const int C0 = t.src_C[0], C1 = t.nsrc > 1 ? t.src_C[1] : 0;
  const int nks0 = (C0 + 31) >> 5, nks = nks0 + ((C1 + 31) >> 5);
  const __amdgpu_buffer_rsrc_t rw = ey_rsrc(t.w, (unsigned)((long)nblk * BN * t.kpad * 2));
  const __amdgpu_buffer_rsrc_t rs0 = ey_rsrc(t.src[0], t.src_lds[0] >= 0 ? 0u : (unsigned)((((long)M - 1) * t.src_cs[0] + C0) * 2));
  const __amdgpu_buffer_rsrc_t rs1 = ey_rsrc(t.nsrc > 1 ? t.src[1] : t.src[0], (t.nsrc < 2 || t.src_lds[1] >= 0) ? 0u : (unsigned)((((long)M - 1) * t.src_cs[1] + C1) * 2));
  bool pv[2];
  int mm[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) { mm[mt] = m0 + mt * 16 + r; pv[mt] = mm[mt] < M; }
  for (int it = wave; it < nitems; it += 4) {
    const int sub = it % nsub, nb = it / nsub;
    unsigned woff[NTI];
#pragma unroll
    for (int nt = 0; nt < NTI; ++nt) woff[nt] = (unsigned)((((nb * BN + (sub * NTI + nt) * 16 + r) * t.kpad) + 8 * g) * 2);
    f32x4 acc[2][NTI];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTI; ++nt) acc[mt][nt] = (f32x4)0.f;
    constexpr int U = NTI >= 4 ? 4 : 8;  // k-steps per batch of loads (U * NTI weight fragments + 2U pixel fragments in flight)
    for (int jb = 0; jb < nks; jb += U) {
      Vec8<f16> af[U][NTI], bf[U][2];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = jb + u;
        const bool on = j < nks, s1 = j >= nks0;                    // (wave-uniform)
        const int c0 = 32 * (s1 ? j - nks0 : j), Cs = s1 ? C1 : C0;
        const int kof = s1 ? C0 + c0 : c0;                          // k offset of this step in the packed row
#pragma unroll
        for (int nt = 0; nt < NTI; ++nt) BufLoad8<f16>::load(af[u][nt], rw, on ? woff[nt] : EY_OOB, kof * 2);
        const bool chan_ok = on && c0 + 8 * g < Cs;                 // channel tail of a source that is not a multiple of 32: zeros
        const int sl = s1 ? t.src_lds[1] : t.src_lds[0], cs = s1 ? t.src_cs[1] : t.src_cs[0];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (sl >= 0) {  // an intermediate of the chain: this tile's rows in LDS
            if (chan_ok) bf[u][mt].load(lds + sl + (mt * 16 + r) * cs + c0 + 8 * g);
            else bf[u][mt].zero();
          } else {
            const unsigned po = (chan_ok && pv[mt]) ? (unsigned)mm[mt] * (unsigned)(cs * 2) + 16u * g : EY_OOB;
            if (s1) BufLoad8<f16>::load(bf[u][mt], rs1, po, c0 * 2);
            else BufLoad8<f16>::load(bf[u][mt], rs0, po, c0 * 2);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (jb + u < nks) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTI; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u][nt].v, bf[u][mt].v, acc[mt][nt], 0, 0, 0);
        }
    }
    const int ch0 = nb * BN + g * 4 * NT + 4 * sub * NTI;
    if (ch0 >= t.Cout) continue;
    BlkEpi e;
    e.bias = t.bias ? t.bias + ch0 : nullptr;
    e.z = t.z ? t.z + ch0 : nullptr;
    e.addz_cs = t.addz_cs; e.Hz = t.Hz; e.Wz = t.Wz; e.act = t.act; e.zsy = t.zsy; e.zsx = t.zsx; e.out_scale = t.out_scale;
    e.y_cs = t.y_cs; e.res_cs = t.res_cs;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (!pv[mt]) continue;
      // LDS-resident tensors are indexed by the pixel's position in the tile, global ones by the pixel
      const int row = mt * 16 + r;
      e.y = (t.y_lds >= 0 ? lds + t.y_lds : t.y) + ch0;
      e.res = t.res ? (t.res_lds >= 0 ? lds + t.res_lds : t.res) + ch0 : nullptr;
      const int oy = mm[mt] / Wo;
      blk_epilogue<NTI>(e, acc[mt], mm[mt], oy, mm[mt] - oy * Wo, t.y_lds >= 0 ? row : -1, (t.res && t.res_lds >= 0) ? row : -1);
    }
  }
}

__global__ __launch_bounds__(256) void block_tile_kernel(const BlkStage* __restrict__ prog, int nstages, BlkExt ext) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TStage* ts = reinterpret_cast<TStage*>(smem);                                   // [TILE_MAX_STAGES]
  f16* lds = reinterpret_cast<f16*>(smem + TILE_MAX_STAGES * sizeof(TStage));     // the tile's chain-internal tensors
  const int b = blockIdx.y, m0 = blockIdx.x * TILE_PX;
  if ((int)threadIdx.x < nstages) {  // resolve every stage once (pointers of this image, LDS placement): no global round trip per stage later
    const BlkStage& s = prog[threadIdx.x];
    TStage t;
    for (int j = 0; j < 2; ++j) {
      const int jj = j < s.nsrc ? j : 0;
      t.src[j] = blk_ptr(ext, s.src[jj], s.src_ext[jj]) + (long)b * s.src_img[jj];
      t.src_lds[j] = s.tile_src_lds[jj]; t.src_cs[j] = s.tile_src_lds[jj] >= 0 ? s.tile_src_lcs[jj] : s.src_cs[jj]; t.src_C[j] = s.src_C[jj];
    }
    t.nsrc = s.nsrc;
    t.w = reinterpret_cast<const f16*>(s.w); t.bias = reinterpret_cast<const float*>(s.bias);
    t.y = const_cast<f16*>(blk_ptr(ext, s.y, s.y_ext)) + (long)b * s.y_img;
    t.y_lds = s.tile_y_lds; t.y_cs = s.tile_y_lds >= 0 ? s.tile_y_lcs : s.y_cs;
    t.res = s.has_res ? blk_ptr(ext, s.res, s.res_ext) + (long)b * s.res_img : nullptr;
    t.res_lds = s.has_res ? s.tile_res_lds : -1; t.res_cs = (s.has_res && s.tile_res_lds >= 0) ? s.tile_res_lcs : s.res_cs;
    t.z = s.has_addz ? blk_ptr(ext, s.addz, s.addz_ext) + (long)b * s.addz_img : nullptr;
    t.kpad = s.kpad; t.nt_pack = s.nt_pack; t.nti = s.tile_nti; t.Cout = s.Cout; t.act = s.act; t.addz_cs = s.addz_cs; t.Hz = s.addz_H; t.Wz = s.addz_W;
    t.zsy = s.zsy; t.zsx = s.zsx; t.out_scale = s.out_scale;
    ts[threadIdx.x] = t;
  }
  const int M = prog[0].Ho * prog[0].Wo, Wo = prog[0].Wo;
  __syncthreads();
  for (int si = 0; si < nstages; ++si) {
    const TStage& t = ts[si];
    switch (t.nti) {
      case 1: tile_conv<1>(t, lds, b, m0, M, Wo); break;
      case 2: tile_conv<2>(t, lds, b, m0, M, Wo); break;
      case 4: tile_conv<4>(t, lds, b, m0, M, Wo); break;
      case 5: tile_conv<5>(t, lds, b, m0, M, Wo); break;
      default: break;
    }
    __syncthreads();  // the tile's stage output (LDS, or global written by this workgroup) is visible to the next stage
  }
}

// ---------------------------------------------------------------------------------------------------------------- host side
static const int kMT[3] = {1, 2, 4};

extern "C" size_t ey_block_program_bytes(int nstages) { return (size_t)(nstages > 0 ? nstages : 0) * sizeof(ey_block_stage); }
extern "C" size_t ey_block_stage_sizeof(void) { return sizeof(ey_block_stage); }

static int blk_view_ok(int64_t addr, int ext, int cs, int C, const char* what, int si) {
  if (ext >= BLK_MAX_EXT) return ey_set_error(EY_EINVAL, "block stage %d: %s: ext index %d out of range", si, what, ext);
  if (ext < 0 && addr == 0) return ey_set_error(EY_EINVAL, "block stage %d: %s: null pointer", si, what);
  if (addr % 16) return ey_set_error(EY_EINVAL, "block stage %d: %s: not 16-byte aligned", si, what);
  if (cs < C || (cs * 2) % 16 || C % 8) return ey_set_error(EY_EINVAL, "block stage %d: %s: cstride %d / channels %d (multiples of 8, 16-byte pixel stride)", si, what, cs, C);
  return EY_OK;
}

// Validates the stages, fills the derived fields (packing geometry, wave-tile shape) and writes the device image of the program
// into out_host (upload it; ey_block_run takes the device copy).
extern "C" int ey_block_compile(const ey_block_stage* st, int nstages, void* out_host, size_t out_bytes) {
  EY_CHECK(st && out_host && nstages > 0, "block_compile: null / empty program");
  EY_CHECK(out_bytes >= ey_block_program_bytes(nstages), "block_compile: output buffer too small");
  ey_block_stage* o = (ey_block_stage*)out_host;
  for (int i = 0; i < nstages; ++i) {
    ey_block_stage s = st[i];
    EY_CHECK(s.H > 0 && s.W > 0 && s.Ho > 0 && s.Wo > 0 && (long)s.H * s.W <= 4096, "block stage %d: extent %dx%d -> %dx%d (maps up to 4096 pixels)", i, s.H, s.W, s.Ho, s.Wo);
    int rc;
    if ((rc = blk_view_ok(s.src[0], s.src_ext[0], s.src_cs[0], s.src_C[0], "src0", i))) return rc;
    EY_CHECK(s.src_img[0] >= 0 && s.y_img >= 0, "block stage %d: negative image stride", i);
    switch (s.op) {
      case EY_BLK_CONV: {
        EY_CHECK((s.k == 1 || s.k == 3) && (s.stride == 1 || s.stride == 2), "block stage %d: conv k=%d stride=%d", i, s.k, s.stride);
        EY_CHECK(s.Ho == (s.H + 2 * (s.k / 2) - s.k) / s.stride + 1 && s.Wo == (s.W + 2 * (s.k / 2) - s.k) / s.stride + 1, "block stage %d: conv output extent", i);
        EY_CHECK(s.nsrc == 1 || s.nsrc == 2, "block stage %d: nsrc=%d", i, s.nsrc);
        if (s.nsrc == 2 && (rc = blk_view_ok(s.src[1], s.src_ext[1], s.src_cs[1], s.src_C[1], "src1", i))) return rc;
        EY_CHECK(s.w && s.Cout > 0 && s.Cout % 8 == 0, "block stage %d: conv Cout=%d (multiple of 8) / weights", i, s.Cout);
        if ((rc = blk_view_ok(s.y, s.y_ext, s.y_cs, s.Cout, "y", i))) return rc;
        if (s.has_res && (rc = blk_view_ok(s.res, s.res_ext, s.res_cs, s.Cout, "res", i))) return rc;
        if (s.has_addz && (rc = blk_view_ok(s.addz, s.addz_ext, s.addz_cs, s.Cout, "addz", i))) return rc;
        EY_CHECK(!s.has_addz || (s.addz_H > 0 && s.addz_W > 0), "block stage %d: addz extent", i);
        EY_CHECK(!s.bias || ey_aligned(s.bias, 16), "block stage %d: bias alignment", i);
        if (s.ngroup < 1) s.ngroup = 1;
        EY_CHECK(s.ngroup == 1 || s.nsrc == 1, "block stage %d: groups need a single source", i);
        EY_CHECK((s.src_g * 2) % 16 == 0 && (s.y_g * 2) % 16 == 0, "block stage %d: group strides", i);
        int Cin = 0;
        for (int j = 0; j < s.nsrc; ++j) Cin += s.src_C[j];
        s.kpad = ey_conv_kpad(s.k * s.k * Cin, 2);
        s.nt_pack = ey_conv_pack_nt(s.Cout);
        s.zsy = s.has_addz ? (float)s.addz_H / (float)s.Ho : 0.f;
        s.zsx = s.has_addz ? (float)s.addz_W / (float)s.Wo : 0.f;
        if (s.w_gmax < 0) s.w_gmax = 0;
        s.lds = 0;
        {  // pointwise-chain form (block_tile_kernel): the 4 waves of a tile's workgroup split nblk * NT / nti items; widest nti with >= 4 items
          const int NT = s.nt_pack, nblk = (s.Cout + 16 * NT - 1) / (16 * NT);
          s.tile_nti = 0;
          if (s.k == 1 && s.stride == 1 && s.ngroup == 1) {
            const int cand[4] = {5, 4, 2, 1};
            for (int q = 0; q < 4 && !s.tile_nti; ++q)
              if (NT % cand[q] == 0 && (cand[q] != 5 || NT == 5) && (nblk * (NT / cand[q]) >= 4 || cand[q] == 1)) s.tile_nti = cand[q];
          }
        }
        // wave tile (MT pixel tiles x NTI row blocks): operands come fragment-wise through the CU's 64 B/clk vector-memory path, so a
        // round of the 16 waves costs ~(MT + NTI) fragment loads per k-step (the MFMAs hide behind them): fewest rounds x loads wins
        const int M = s.Ho * s.Wo, NT = s.nt_pack, nblk = (s.Cout + 16 * NT - 1) / (16 * NT);
        long best = -1;
        for (int a = 0; a < 3; ++a) {
          for (int nti = 1; nti <= 5; ++nti) {
            const int mt = kMT[a];
            if (NT % nti || nti == 3 || mt * nti > 8 || (nti == 5 && mt > 1)) continue;  // (the instantiated wave tiles: <= 8 accumulator blocks, 128 VGPRs)
            const long items = (long)s.ngroup * ((M + 16 * mt - 1) / (16 * mt)) * nblk * (NT / nti);
            const long rounds = (items + BLK_WAVES - 1) / BLK_WAVES;
            const long cost = rounds * (mt + nti) * 16 + mt * nti;
            if (best < 0 || cost < best) { best = cost; s.mt = mt; s.nti = nti; }
          }
        }
        break;
      }
      case EY_BLK_DW:
        EY_CHECK((s.k == 3 || s.k == 5 || s.k == 7) && s.w && s.Ho == s.H && s.Wo == s.W, "block stage %d: depthwise k=%d", i, s.k);
        if ((rc = blk_view_ok(s.y, s.y_ext, s.y_cs, s.src_C[0], "y", i))) return rc;
        EY_CHECK(ey_aligned(s.w, 16), "block stage %d: depthwise weights alignment", i);
        break;
      case EY_BLK_DWT:
        EY_CHECK(s.H >= 2 && s.W >= 2 && s.Ho == s.H / 2 && s.Wo == s.W / 2, "block stage %d: dwt extent", i);
        if ((rc = blk_view_ok(s.y, s.y_ext, s.y_cs, 4 * s.src_C[0], "y", i))) return rc;
        break;
      case EY_BLK_POOL:
        EY_CHECK(s.Ho == s.H && s.Wo == s.W && (long)s.H * s.W * 16 * 2 <= BLK_LDS_BYTES, "block stage %d: pool map %dx%d does not fit LDS", i, s.H, s.W);
        if ((rc = blk_view_ok(s.y, s.y_ext, s.y_cs, 3 * s.src_C[0], "y", i))) return rc;
        break;
      case EY_BLK_LINATTN:
        EY_CHECK(s.heads > 0 && s.heads % 2 == 0 && s.Cout == 64 * s.heads && s.src_C[0] == 3 * s.Cout, "block stage %d: linear attention needs an even number of 64-channel heads", i);
        EY_CHECK(2 * sizeof(LinAttnLds) <= BLK_LDS_BYTES, "block stage %d: LDS", i);
        if ((rc = blk_view_ok(s.y, s.y_ext, s.y_cs, s.Cout, "y", i))) return rc;
        break;
      default:
        return ey_set_error(EY_EINVAL, "block stage %d: unknown op %d", i, s.op);
    }
    o[i] = s;
  }
  // ---- pointwise-chain lowering (block_tile_kernel): tensors that only the chain touches (absolute refs produced by a stage) live in
  // LDS, one row of C + 8 halves per pixel of the 32-pixel tile; everything else stays in global memory
  {
    struct Own { int64_t addr; int cs, C, lds, lcs; };
    Own own[TILE_MAX_STAGES];
    int nown = 0;
    long top = 0;
    bool ok = nstages <= TILE_MAX_STAGES;
    for (int i = 0; i < nstages && ok; ++i) {
      ey_block_stage& s = o[i];
      if (s.op != EY_BLK_CONV || s.tile_nti == 0) { ok = false; break; }
      auto find = [&](int64_t addr, int ext, int cs, int& lds, int& lcs) {
        lds = -1; lcs = 0;
        if (ext >= 0) return;
        for (int q = 0; q < nown; ++q)
          if (own[q].cs == cs && addr >= own[q].addr && addr < own[q].addr + (int64_t)own[q].C * 2) {
            lds = own[q].lds + (int)((addr - own[q].addr) / 2);
            lcs = own[q].lcs;
          }
      };
      for (int j = 0; j < 2; ++j) {
        s.tile_src_lds[j] = -1; s.tile_src_lcs[j] = 0;
        if (j < s.nsrc) find(s.src[j], s.src_ext[j], s.src_cs[j], s.tile_src_lds[j], s.tile_src_lcs[j]);
      }
      s.tile_res_lds = -1; s.tile_res_lcs = 0;
      if (s.has_res) find(s.res, s.res_ext, s.res_cs, s.tile_res_lds, s.tile_res_lcs);
      s.tile_y_lds = -1; s.tile_y_lcs = 0;
      if (s.y_ext < 0) {  // an intermediate the caller never sees
        int lds, lcs;
        find(s.y, s.y_ext, s.y_cs, lds, lcs);
        if (lds >= 0) { s.tile_y_lds = lds; s.tile_y_lcs = lcs; }  // (written into a slice of an earlier intermediate)
        else {
          Own w; w.addr = s.y; w.cs = s.y_cs; w.C = s.Cout; w.lcs = s.Cout + 8; w.lds = (int)top;
          top += (long)TILE_PX * w.lcs;
          own[nown++] = w;
          s.tile_y_lds = w.lds; s.tile_y_lcs = w.lcs;
        }
      }
    }
    o[0].tile_lds_bytes = (ok && top * 2 <= 56 * 1024) ? (int32_t)(top * 2) : -1;
  }
  return EY_OK;
}

static int block_launch(const void* program_dev, int nstages, int B, const void* const* ext_ptrs_host, int next, long long* tstamps, ey_stream_t stream) {
  EY_CHECK(program_dev && nstages > 0 && B > 0, "block_run: null / empty program");
  EY_CHECK(next >= 0 && next <= BLK_MAX_EXT && (next == 0 || ext_ptrs_host), "block_run: %d external tensors (0..%d)", next, BLK_MAX_EXT);
  BlkExt ext;
  for (int i = 0; i < BLK_MAX_EXT; ++i) {
    ext.p[i] = i < next ? (char*)const_cast<void*>(ext_ptrs_host[i]) : nullptr;
    EY_CHECK(i >= next || (ext.p[i] && ey_aligned(ext.p[i], 16)), "block_run: external tensor %d null / not 16-byte aligned", i);
  }
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BLK_LDS_BYTES) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "block_run: cannot reserve %d B of LDS", BLK_LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(block_kernel, dim3(B), dim3(BLK_THREADS), BLK_LDS_BYTES, (hipStream_t)stream, (const BlkStage*)program_dev, nstages, ext, tstamps);
  EY_LAUNCH_CHECK("ey_block_run");
  return EY_OK;
}

extern "C" int ey_block_run(const void* program_dev, int nstages, int B, const void* const* ext_ptrs_host, int next, ey_stream_t stream) {
  return block_launch(program_dev, nstages, B, ext_ptrs_host, next, nullptr, stream);
}

// Pointwise-chain form: every stage must be tile-executable (ey_block_tileable on the compiled program) and share the map H x W.
extern "C" int ey_block_tileable(const ey_block_stage* compiled_host, int nstages) {
  if (!compiled_host || nstages <= 0 || nstages > TILE_MAX_STAGES || compiled_host[0].tile_lds_bytes < 0) return 0;
  for (int i = 0; i < nstages; ++i) {
    const ey_block_stage& s = compiled_host[i];
    if (s.op != EY_BLK_CONV || s.tile_nti == 0 || s.Ho != compiled_host[0].Ho || s.Wo != compiled_host[0].Wo || s.H != s.Ho || s.W != s.Wo) return 0;
  }
  return 1;
}
extern "C" int ey_block_run_tiles(const void* program_dev, int nstages, int B, int H, int W, int lds_bytes, const void* const* ext_ptrs_host, int next,
                                  ey_stream_t stream) {
  EY_CHECK(program_dev && nstages > 0 && nstages <= TILE_MAX_STAGES && B > 0 && H > 0 && W > 0 && lds_bytes >= 0 && lds_bytes <= 56 * 1024,
           "block_run_tiles: bad program (%d stages, %d B of LDS)", nstages, lds_bytes);
  EY_CHECK(next >= 0 && next <= BLK_MAX_EXT && (next == 0 || ext_ptrs_host), "block_run_tiles: %d external tensors (0..%d)", next, BLK_MAX_EXT);
  BlkExt ext;
  for (int i = 0; i < BLK_MAX_EXT; ++i) {
    ext.p[i] = i < next ? (char*)const_cast<void*>(ext_ptrs_host[i]) : nullptr;
    EY_CHECK(i >= next || (ext.p[i] && ey_aligned(ext.p[i], 16)), "block_run_tiles: external tensor %d null / not 16-byte aligned", i);
  }
  const int mtiles = (H * W + TILE_PX - 1) / TILE_PX;
  EY_CHECK(B <= 65535, "block_run_tiles: batch %d", B);
  const size_t lds = TILE_MAX_STAGES * sizeof(TStage) + (size_t)lds_bytes;
  hipLaunchKernelGGL(block_tile_kernel, dim3((unsigned)mtiles, (unsigned)B), dim3(256), lds, (hipStream_t)stream, (const BlkStage*)program_dev, nstages, ext);
  EY_LAUNCH_CHECK("ey_block_run_tiles");
  return EY_OK;
}
// developer tool: same launch; workgroup 0 also writes wall_clock64() (100 MHz) at the start and after every stage into tstamps_dev[nstages + 1]
extern "C" int ey_block_run_timed(const void* program_dev, int nstages, int B, const void* const* ext_ptrs_host, int next, long long* tstamps_dev, ey_stream_t stream) {
  EY_CHECK(tstamps_dev, "block_run_timed: null timestamp buffer");
  return block_launch(program_dev, nstages, B, ext_ptrs_host, next, tstamps_dev, stream);
}
