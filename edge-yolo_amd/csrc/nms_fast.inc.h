// K11 fast path (predict mode, single label): NMS as THREE short kernels that use the whole chip, instead of one 1024-thread
// workgroup per image running ~45 barrier-separated sequential rounds (nms_select_greedy_kernel, which stays as the general path:
// validation mode, top-k mode, and the fallback below).  Included by head_nms.hip (shares iou arithmetic, key layout, sort helpers).
//
//   nf_select  (1 workgroup / image)   the K best candidate keys of the image (K <= 2048; every candidate when there are fewer):
//                                      radix select on the score bits over an adaptive 12-bit digit, register bitonic sort, then the
//                                      candidates' class-offset boxes + areas in score order -> scratch
//   nf_mask    (144 x B workgroups)    the upper-triangular suppression matrix of those K candidates as 64 x 64 bit tiles:
//                                      bit i of word (cb, rb)[lane] = "candidate rb*64+i (earlier) overlaps candidate cb*64+lane beyond
//                                      iou_thres" -- torchvision's IoU expression op by op (iou_gt_fast: the division is only executed
//                                      when a reciprocal estimate lands within 1e-6 of the threshold, so decisions are bit-identical)
//   nf_resolve (1 workgroup / image)   greedy = "keep j iff no KEPT earlier candidate overlaps it": column blocks of 64 in order; per
//                                      block mask arithmetic only (words AND kept-words of the earlier blocks, then the in-wave fixed
//                                      point of nms_resolve on the diagonal tile); writes rows / count / indices
//
// Exactness: the K selected keys are exactly the K best (all keys >= a threshold), the matrix holds every pair among them, and the
// resolve visits them in score order, so the kept set equals the sequential greedy's as long as max_det boxes are found among the K or
// the K are all there is.  Otherwise (dense scenes that keep fewer than max_det of the best K, huge score-tie groups) the image's `done`
// flag stays 0 and nms_select_greedy_kernel, launched right behind with that flag array, redoes exactly those images (others exit at
// once).  Cost on the random-init model's output (8400 candidates, 300 kept at rank ~1200): 240 us -> see profiles/r03_*.
#define NF_KMAX 2048
#define NF_NB (NF_KMAX / 64)
#define NF_TILES (NF_NB * (NF_NB + 1) / 2)
#define NF_GROUPS 144  // sum over cb of ceil((cb+1)/4): (column block, 4 row blocks) work items of nf_mask per image

struct NfMeta { int n_sel, n_total, done, pad; };
static inline size_t nf_image_bytes() {
  const size_t b = (size_t)NF_KMAX * (8 + 16 + 4) + (size_t)NF_TILES * 64 * 8 + sizeof(NfMeta);
  return (b + 255) & ~(size_t)255;
}
__device__ __forceinline__ unsigned long long* nf_skey(char* img) { return reinterpret_cast<unsigned long long*>(img); }
__device__ __forceinline__ float4* nf_cbox(char* img) { return reinterpret_cast<float4*>(img + (size_t)NF_KMAX * 8); }
__device__ __forceinline__ float* nf_area(char* img) { return reinterpret_cast<float*>(img + (size_t)NF_KMAX * 24); }
__device__ __forceinline__ unsigned long long* nf_mask(char* img) { return reinterpret_cast<unsigned long long*>(img + (size_t)NF_KMAX * 28); }
__device__ __forceinline__ NfMeta* nf_meta(char* img) { return reinterpret_cast<NfMeta*>(img + (size_t)NF_KMAX * 28 + (size_t)NF_TILES * 512); }

// iou_gt with the division taken only when it can matter.  q = inter * rcp(union) is within ~2 ulp (2.4e-7 relative) of the exact
// quotient and the correctly rounded quotient within 0.5 ulp of it, so outside [thr_lo, thr_hi] = thr * (1 -+ 1e-6) the comparison
// `fl(inter / union) > thr` is already decided; inside (and for NaN / zero / denormal unions, where rcp is not trusted) the exact
// expression of iou_gt runs.  thr_hi is clamped to >= 1e-30 so that an underflowing quotient is never called positive.
__device__ __forceinline__ bool iou_gt_fast(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1, float bx2, float by2, float barea, float thr,
                                            float thr_lo, float thr_hi) {
  const float w = __fsub_rn(fminf(ax2, bx2), fmaxf(ax1, bx1));
  const float h = __fsub_rn(fminf(ay2, by2), fmaxf(ay1, by1));
  if (!(w > 0.f) || !(h > 0.f)) {
    if (w != w || h != h) {
      const float ww = fmaxf(0.f, w), hh = fmaxf(0.f, h);
      const float inter = __fmul_rn(ww, hh);
      return __fdiv_rn(inter, __fsub_rn(__fadd_rn(aarea, barea), inter)) > thr;
    }
    return false;
  }
  const float inter = __fmul_rn(w, h);
  const float uni = __fsub_rn(__fadd_rn(aarea, barea), inter);
  const float q = __fmul_rn(inter, __builtin_amdgcn_rcpf(uni));
  if (q > thr_hi && uni > 1e-30f) return true;
  if (q < thr_lo) return false;
  return __fdiv_rn(inter, uni) > thr;
}

// Descending bitonic sort of 2048 keys, two per thread (elements 2t and 2t+1): partner distance 1 is inside the thread, 2..64 a lane
// shuffle (thread distance < 64), >= 128 through LDS (10 of the 66 steps).  Keys are unique or 0 (padding).
__device__ __forceinline__ void nf_sort2048(unsigned long long& k0, unsigned long long& k1, unsigned long long* xch) {
  const int t = threadIdx.x;
#pragma unroll 1
  for (int k = 2; k <= 2048; k <<= 1) {
#pragma unroll 1
    for (int j = k >> 1; j > 0; j >>= 1) {
      const bool desc = ((2 * t) & k) == 0;  // (k >= 2: both elements of a thread lie in the same k-block)
      if (j == 1) {
        const unsigned long long mx = k0 > k1 ? k0 : k1, mn = k0 > k1 ? k1 : k0;
        k0 = desc ? mx : mn;
        k1 = desc ? mn : mx;
        continue;
      }
      unsigned long long o0, o1;
      const int tj = j >> 1;  // partner thread distance
      if (tj >= 64) {
        xch[2 * t] = k0;
        xch[2 * t + 1] = k1;
        __syncthreads();
        o0 = xch[2 * (t ^ tj)];
        o1 = xch[2 * (t ^ tj) + 1];
        __syncthreads();
      } else {
        o0 = ((unsigned long long)__shfl_xor((unsigned)(k0 >> 32), tj, 64) << 32) | __shfl_xor((unsigned)k0, tj, 64);
        o1 = ((unsigned long long)__shfl_xor((unsigned)(k1 >> 32), tj, 64) << 32) | __shfl_xor((unsigned)k1, tj, 64);
      }
      const bool keep_max = desc == ((t & tj) == 0);
      k0 = keep_max ? (k0 > o0 ? k0 : o0) : (k0 < o0 ? k0 : o0);
      k1 = keep_max ? (k1 > o1 ? k1 : o1) : (k1 < o1 ? k1 : o1);
    }
  }
}

struct NfSelShared {
  unsigned hist[4096];
  unsigned long long sel[NF_KMAX];
  unsigned long long xch[NF_KMAX];
  unsigned wred[16][3];
  unsigned vmin, vmax, total;
  int bsel, cnt;
};

__global__ __launch_bounds__(1024) void nf_select_kernel(int nc, int A, long P, long nkeys, const float* __restrict__ boxsrc, long img_stride, int max_nms, float max_wh,
                                                         int agnostic, int K, const unsigned long long* __restrict__ gkeys, const int* __restrict__ cls_id,
                                                         char* __restrict__ scratch, size_t img_bytes) {
  __shared__ NfSelShared S;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long* keys = gkeys + (long)b * P;
  const float* pb = boxsrc + (long)b * img_stride;
  const int* cid = cls_id + (long)b * P;
  char* img = scratch + (size_t)b * img_bytes;

  // ---- pass 0: number of candidates, range of their score bits
  unsigned n = 0, vmin = 0xFFFFFFFFu, vmax = 0u;
  for (long i = tid; i < nkeys; i += 1024) {
    const unsigned long long k = keys[i];
    if (k != 0ull) {
      const unsigned v = (unsigned)(k >> 32);
      ++n;
      vmin = min(vmin, v);
      vmax = max(vmax, v);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n += __shfl_xor(n, off, 64);
    vmin = min(vmin, (unsigned)__shfl_xor(vmin, off, 64));
    vmax = max(vmax, (unsigned)__shfl_xor(vmax, off, 64));
  }
  if (lane == 0) { S.wred[wave][0] = n; S.wred[wave][1] = vmin; S.wred[wave][2] = vmax; }
  __syncthreads();
  if (tid == 0) {
    unsigned t = 0, mn = 0xFFFFFFFFu, mx = 0u;
    for (int w = 0; w < 16; ++w) { t += S.wred[w][0]; mn = min(mn, S.wred[w][1]); mx = max(mx, S.wred[w][2]); }
    S.total = t; S.vmin = mn; S.vmax = mx; S.cnt = 0;
  }
  __syncthreads();
  const unsigned total = S.total;
  // ---- threshold T on the score bits: keys with v >= T are exactly the best n_sel <= K candidates
  unsigned T = 0u;  // (every candidate)
  if (total > (unsigned)K) {
    unsigned lo_r = S.vmin, hi_r = S.vmax, Q = (unsigned)K;
    T = hi_r + 1u;  // nothing selected yet (vmax < 0xFFFFFFFF: scores are finite positive floats)
    while (true) {  // wave-uniform state; at most three rounds for a 32-bit range
      const unsigned range = hi_r - lo_r;
      int shift = 0;
      while ((range >> shift) >= 4096u) ++shift;
      for (int i = tid; i < 4096; i += 1024) S.hist[i] = 0u;
      if (tid == 0) S.bsel = 4096;
      __syncthreads();
      for (long i = tid; i < nkeys; i += 1024) {
        const unsigned long long k = keys[i];
        if (k != 0ull) {
          const unsigned v = (unsigned)(k >> 32);
          if (v >= lo_r && v <= hi_r) atomicAdd(&S.hist[(v - lo_r) >> shift], 1u);
        }
      }
      __syncthreads();
      nms_block_suffix_scan(S.hist);  // hist[d] = #keys of the range with digit >= d
      {
        int best = 4096;  // smallest digit whose suffix count fits the quota (counts fall with d: the first hit of a thread is its smallest)
        for (int d = 4 * tid + 3; d >= 4 * tid; --d)
          if (S.hist[d] <= Q) best = d;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
        if (lane == 0 && best < 4096) atomicMin(&S.bsel, best);
      }
      __syncthreads();
      const int bs = S.bsel;
      const unsigned taken = bs < 4096 ? S.hist[bs] : 0u;
      __syncthreads();  // (hist is rewritten by the next round)
      if (bs < 4096) T = lo_r + ((unsigned)bs << shift);
      Q -= taken;
      if (bs == 0 || shift == 0 || Q == 0u) break;
      const unsigned nlo = lo_r + ((unsigned)(bs - 1) << shift);  // the boundary digit: more keys than the quota left
      hi_r = min(hi_r, nlo + ((1u << shift) - 1u));
      lo_r = nlo;
    }
  }
  // ---- gather (unordered) + sort
  for (long i = tid; i < nkeys; i += 1024) {
    const unsigned long long k = keys[i];
    if (k != 0ull && (unsigned)(k >> 32) >= T) {
      const int p = atomicAdd(&S.cnt, 1);
      if (p < NF_KMAX) S.sel[p] = k;
    }
  }
  __syncthreads();
  const int nsel = min(S.cnt, K);  // (== S.cnt by construction)
  unsigned long long k0, k1;
  if (nsel <= 1024) {
    k0 = nms_sort1024(tid < nsel ? S.sel[tid] : 0ull, S.xch);
    __syncthreads();
    S.sel[tid] = k0;
  } else {
    k0 = 2 * tid < nsel ? S.sel[2 * tid] : 0ull;
    k1 = 2 * tid + 1 < nsel ? S.sel[2 * tid + 1] : 0ull;
    __syncthreads();
    nf_sort2048(k0, k1, S.xch);
    S.sel[2 * tid] = k0;
    S.sel[2 * tid + 1] = k1;
  }
  __syncthreads();
  // ---- candidate records in score order: key, class-offset box (ops.py:289), area -- the operands of torchvision's nms
  for (int e = tid; e < nsel; e += 1024) {
    const unsigned long long key = S.sel[e];
    const int a = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
    const float cx = pb[a], cy = pb[(long)A + a], w = pb[2L * A + a], h = pb[3L * A + a];
    const float hw = __fmul_rn(w, 0.5f), hh = __fmul_rn(h, 0.5f);  // xywh2xyxy, ops.py:430-432
    const float ux1 = __fsub_rn(cx, hw), uy1 = __fsub_rn(cy, hh), ux2 = __fadd_rn(cx, hw), uy2 = __fadd_rn(cy, hh);
    const float off = agnostic ? 0.f : __fmul_rn((float)cid[a], max_wh);
    const float x1 = __fadd_rn(ux1, off), y1 = __fadd_rn(uy1, off), x2 = __fadd_rn(ux2, off), y2 = __fadd_rn(uy2, off);
    nf_skey(img)[e] = key;
    nf_cbox(img)[e] = make_float4(x1, y1, x2, y2);
    nf_area(img)[e] = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
  }
  if (tid == 0) {
    NfMeta* m = nf_meta(img);
    m->n_sel = nsel;
    m->n_total = (int)total;
    m->done = 0;
  }
}

// (column block cb, row blocks 4*grp .. 4*grp+3): one wave per 64 x 64 tile.
__global__ __launch_bounds__(256) void nf_mask_kernel(float thr, float thr_lo, float thr_hi, char* __restrict__ scratch, size_t img_bytes) {
  __shared__ KeptBox rows[4][64];
  char* img = scratch + (size_t)blockIdx.y * img_bytes;
  const int nsel = nf_meta(img)->n_sel;
  int cb = 0, g = (int)blockIdx.x;
  while (g >= cb / 4 + 1) { g -= cb / 4 + 1; ++cb; }  // (<= 32 scalar steps)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = 4 * g + wave;
  if (cb * 64 >= nsel || rb > cb) return;  // (no workgroup barrier below: waves are independent)
  const int j = cb * 64 + lane, i0 = rb * 64;
  const bool valid = j < nsel;
  const float4 me = valid ? nf_cbox(img)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float marea = valid ? nf_area(img)[j] : 0.f;
  {
    const int i = i0 + lane;  // (i0 + 63 < nsel whenever rb < cb; the diagonal tile masks its tail through `valid` and i < lane)
    const float4 rbx = i < nsel ? nf_cbox(img)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    rows[wave][lane] = KeptBox{rbx.x, rbx.y, rbx.z, rbx.w, i < nsel ? nf_area(img)[i] : 0.f};
  }
  __builtin_amdgcn_wave_barrier();
  const int nrow = min(64, nsel - i0);
  const int lim = rb == cb ? lane : 64;  // the diagonal tile: only earlier candidates
  unsigned long long word = 0ull;
#pragma unroll 4
  for (int i = 0; i < nrow; ++i) {
    const KeptBox r = rows[wave][i];
    if (i < lim && iou_gt_fast(r.x1, r.y1, r.x2, r.y2, r.area, me.x, me.y, me.z, me.w, marea, thr, thr_lo, thr_hi)) word |= 1ull << i;
  }
  nf_mask(img)[((size_t)(cb * (cb + 1) / 2 + rb)) * 64 + lane] = valid ? word : 0ull;
}

struct NfResShared {
  unsigned long long buf[8][NF_NB][64];  // the mask words of 8 column blocks (128 KB)
  unsigned long long kw[NF_NB];          // kept bits per column block
  int kpre[NF_NB + 1];                   // kept boxes before each block
  int nkept, stop;
};

__global__ __launch_bounds__(1024) void nf_resolve_kernel(int nc, int A, long P, const float* __restrict__ boxsrc, long img_stride, int max_det, int max_nms,
                                                          const int* __restrict__ cls_id, char* __restrict__ scratch, size_t img_bytes, float* __restrict__ out_boxes,
                                                          int* __restrict__ out_count, int* __restrict__ out_index) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  NfResShared& S = *reinterpret_cast<NfResShared*>(smem);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* img = scratch + (size_t)b * img_bytes;
  NfMeta* meta = nf_meta(img);
  const int nsel = meta->n_sel, ntotal = meta->n_total;
  const int NB = (nsel + 63) >> 6;
  const unsigned long long* mask = nf_mask(img);
  if (tid < NF_NB) S.kw[tid] = 0ull;
  if (tid == 0) { S.nkept = 0; S.stop = 0; }
  __syncthreads();
  for (int cb0 = 0; cb0 < NB; cb0 += 8) {
    const int ncb = min(8, NB - cb0);
    for (int c = 0; c < ncb; ++c)
      for (int rb = wave; rb <= cb0 + c; rb += 16) S.buf[c][rb][lane] = mask[((size_t)((cb0 + c) * (cb0 + c + 1) / 2 + rb)) * 64 + lane];
    __syncthreads();
    if (wave == 0) {  // the sequential part: one wave, no barrier inside (it reads kept-words it wrote itself)
      int nk = S.nkept;
      for (int c = 0; c < ncb && nk < max_det; ++c) {
        const int cb = cb0 + c;
        unsigned long long hit = 0ull;
#pragma unroll 4
        for (int rb = 0; rb < cb; ++rb) hit |= S.buf[c][rb][lane] & S.kw[rb];
        const bool alive = (cb * 64 + lane < nsel) && hit == 0ull;
        unsigned long long km = nms_resolve(S.buf[c][cb][lane], alive);
        const int room = max_det - nk;
        if (__popcll(km) > room) {  // keep the first `room` of them (earlier candidates first); nothing after matters
          unsigned long long t = km;
          for (int r = 0; r < room; ++r) t &= t - 1ull;  // clear the lowest `room` bits -> what is left are the surplus bits
          km ^= t;
        }
        if (lane == 0) S.kw[cb] = km;  // (LDS serves a wave's accesses in order: the next block's reads see it)
        nk += __popcll(km);
        __builtin_amdgcn_wave_barrier();
      }
      if (lane == 0) { S.nkept = nk; S.stop = nk >= max_det; }
    }
    __syncthreads();
    if (S.stop) break;
  }
  // kept boxes before each block (kw = 0 for blocks the loop never reached)
  if (tid == 0) {
    int run = 0;
    for (int cb = 0; cb < NF_NB; ++cb) { S.kpre[cb] = run; run += __popcll(S.kw[cb]); }
    S.kpre[NF_NB] = run;
  }
  __syncthreads();
  const int nk = min(S.nkept, max_det);
  const float* pb = boxsrc + (long)b * img_stride;
  const int* cid = cls_id + (long)b * P;
  float* ob = out_boxes + (long)b * max_det * 6;
  for (int e = tid; e < nsel; e += 1024) {
    const unsigned long long w = S.kw[e >> 6];
    if ((w >> (e & 63)) & 1ull) {
      const int rank = S.kpre[e >> 6] + __popcll(w & ((1ull << (e & 63)) - 1ull));
      const unsigned long long key = nf_skey(img)[e];
      const int a = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
      const float cx = pb[a], cy = pb[(long)A + a], ww = pb[2L * A + a], hh = pb[3L * A + a];
      const float hw = __fmul_rn(ww, 0.5f), hh2 = __fmul_rn(hh, 0.5f);
      float* o = ob + rank * 6;
      o[0] = __fsub_rn(cx, hw); o[1] = __fsub_rn(cy, hh2); o[2] = __fadd_rn(cx, hw); o[3] = __fadd_rn(cy, hh2);
      o[4] = __uint_as_float((unsigned)(key >> 32));
      o[5] = (float)cid[a];
      if (out_index) out_index[(long)b * max_det + rank] = a;
    }
  }
  for (int r = nk * 6 + tid; r < max_det * 6; r += 1024) ob[r] = 0.f;
  if (out_index)
    for (int r = nk + tid; r < max_det; r += 1024) out_index[(long)b * max_det + r] = -1;
  if (tid == 0) {
    out_count[b] = nk;
    // complete when max_det boxes were found or every candidate the reference would look at (all of them, capped at max_nms) was among the K
    meta->done = (nk >= max_det || nsel == ntotal || nsel >= max_nms) ? 1 : 0;
  }
}
