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

// Descending bitonic sort of n2 <= 1024 keys (n2 a power of two >= 64), one per thread; threads >= n2 carry zeros through the same
// steps (their exchanges stay among themselves).  n2 = 64 needs no barrier at all.
__device__ __forceinline__ unsigned long long nf_sort_n(unsigned long long key, unsigned long long* xch, int n2) {
  const int t = threadIdx.x;
#pragma unroll 1
  for (int k = 2; k <= n2; k <<= 1) {
    const bool desc = (t & k) == 0;
#pragma unroll 1
    for (int j = k >> 1; j > 0; j >>= 1) {
      unsigned long long other;
      if (j >= 64) {
        xch[t] = key;
        __syncthreads();
        other = xch[t ^ j];
        __syncthreads();
      } else {
        const unsigned lo = __shfl_xor((unsigned)key, j, 64), hi = __shfl_xor((unsigned)(key >> 32), j, 64);
        other = ((unsigned long long)hi << 32) | lo;
      }
      const bool keep_max = desc == ((t & j) == 0);
      key = keep_max ? (key > other ? key : other) : (key < other ? key : other);
    }
  }
  return key;
}

struct NfSelShared {
  unsigned hist[4096];                // counts per score digit, then suffix counts c(d) = #keys with digit >= d
  unsigned long long sel[NF_KMAX];    // selected keys, grouped by digit (descending), unordered inside a digit
  unsigned long long xch[NF_KMAX];    // fill counters of the scatter (as unsigned[4096]) / exchange buffer of the bitonic fallback / sorted keys
  unsigned wred[16][3];
  unsigned vmin, vmax, total, maxbin;
  int bsel;
};

// KPT > 0: the image's keys live in registers (KPT per thread, nkeys <= 1024 * KPT) and are read from memory once;
// KPT == 0: any nkeys, every pass re-reads them (L2).
//
// Selection and sort share ONE 4096-bin histogram of the score bits over [vmin, vmax] (digit = (v - vmin) >> shift): the suffix counts
// c(d) give (a) the smallest digit whose keys-and-above fit K -- the selection (whole digits only: a little less than K when the
// boundary digit is crowded; what is left out is the general kernel's business if it ever matters) -- and (b) every selected key's
// position up to the order inside its digit: pos = c(d + 1) + (rank among the ~2 keys of digit d).  That replaces the 66-step bitonic
// sort of 2048 keys by a scatter and a few compares; digits holding more than 64 selected keys (score ties, clustered scores) fall back
// to the bitonic network.
template <int KPT>
__global__ __launch_bounds__(1024) void nf_select_kernel(int nc, int A, long P, long nkeys, const float* __restrict__ boxsrc, long img_stride, int max_nms, float max_wh,
                                                         int agnostic, int K, const unsigned long long* __restrict__ gkeys, const int* __restrict__ cls_id,
                                                         char* __restrict__ scratch, size_t img_bytes) {
  __shared__ NfSelShared S;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long* keys = gkeys + (long)b * P;
  const float* pb = boxsrc + (long)b * img_stride;
  const int* cid = cls_id + (long)b * P;
  char* img = scratch + (size_t)b * img_bytes;
  unsigned long long kr[KPT > 0 ? KPT : 1];
  if constexpr (KPT > 0) {
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      const long i = tid + 1024L * q;
      kr[q] = i < nkeys ? keys[i] : 0ull;
    }
  }
  auto for_each_key = [&](auto&& f) {
    if constexpr (KPT > 0) {
#pragma unroll
      for (int q = 0; q < KPT; ++q) f(kr[q]);
    } else {
      for (long i = tid; i < nkeys; i += 1024) f(keys[i]);
    }
  };

  // ---- pass 0: number of candidates, range of their score bits
  unsigned n = 0, vmin = 0xFFFFFFFFu, vmax = 0u;
  for_each_key([&](unsigned long long k) {
    if (k != 0ull) {
      const unsigned v = (unsigned)(k >> 32);
      ++n;
      vmin = min(vmin, v);
      vmax = max(vmax, v);
    }
  });
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n += __shfl_xor(n, off, 64);
    vmin = min(vmin, (unsigned)__shfl_xor(vmin, off, 64));
    vmax = max(vmax, (unsigned)__shfl_xor(vmax, off, 64));
  }
  if (lane == 0) { S.wred[wave][0] = n; S.wred[wave][1] = vmin; S.wred[wave][2] = vmax; }
  for (int i = tid; i < 4096; i += 1024) { S.hist[i] = 0u; reinterpret_cast<unsigned*>(S.xch)[i] = 0u; }
  __syncthreads();
  if (tid == 0) {
    unsigned t = 0, mn = 0xFFFFFFFFu, mx = 0u;
    for (int w = 0; w < 16; ++w) { t += S.wred[w][0]; mn = min(mn, S.wred[w][1]); mx = max(mx, S.wred[w][2]); }
    S.total = t; S.vmin = mn; S.vmax = mx; S.bsel = 4096; S.maxbin = 0u;
  }
  __syncthreads();
  const unsigned total = S.total;
  if (total == 0u) {  // nothing passed conf: no records, the resolve kernel writes the empty result
    if (tid == 0) { NfMeta* m = nf_meta(img); m->n_sel = 0; m->n_total = 0; m->done = 0; }
    return;
  }
  const unsigned lo_r = S.vmin;
  int shift = 0;
  while (((S.vmax - lo_r) >> shift) >= 4096u) ++shift;
  for_each_key([&](unsigned long long k) {
    if (k != 0ull) atomicAdd(&S.hist[((unsigned)(k >> 32) - lo_r) >> shift], 1u);
  });
  __syncthreads();
  {
    unsigned mb = max(max(S.hist[4 * tid], S.hist[4 * tid + 1]), max(S.hist[4 * tid + 2], S.hist[4 * tid + 3]));  // (bins above the selection included: harmless)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mb = max(mb, (unsigned)__shfl_xor(mb, off, 64));
    if (lane == 0) atomicMax(&S.maxbin, mb);
  }
  nms_block_suffix_scan(S.hist);  // hist[d] = c(d) = #keys with digit >= d  (2 barriers inside: maxbin is complete afterwards)
  int bs = 0;  // first selected digit
  if (total > (unsigned)K) {
    int best = 4096;  // smallest digit whose suffix count fits K (counts fall with d: a thread's first hit from above is its smallest)
    for (int d = 4 * tid + 3; d >= 4 * tid; --d)
      if (S.hist[d] <= (unsigned)K) best = d;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
    if (lane == 0 && best < 4096) atomicMin(&S.bsel, best);
    __syncthreads();
    bs = S.bsel;
  }
  const int nsel = bs < 4096 ? (int)S.hist[bs] : 0;
  // ---- scatter by digit: pos = c(d + 1) + arrival order inside the digit
  unsigned* fill = reinterpret_cast<unsigned*>(S.xch);
  for_each_key([&](unsigned long long k) {
    if (k != 0ull) {
      const int d = (int)(((unsigned)(k >> 32) - lo_r) >> shift);
      if (d >= bs) S.sel[(d < 4095 ? S.hist[d + 1] : 0u) + atomicAdd(&fill[d], 1u)] = k;
    }
  });
  __syncthreads();
  const bool crowded = S.maxbin > 64u;  // (block-uniform)
  if (!crowded) {
    // order inside a digit: rank = number of larger keys among its (few) keys
    unsigned long long mine[2];
    int dst[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 1024 * q;
      dst[q] = -1;
      if (e < nsel) {
        const unsigned long long k = S.sel[e];
        const int d = (int)(((unsigned)(k >> 32) - lo_r) >> shift);
        const int s0 = d < 4095 ? (int)S.hist[d + 1] : 0, s1 = (int)S.hist[d];
        int r = 0;
        for (int x = s0; x < s1; ++x) r += S.sel[x] > k;
        mine[q] = k;
        dst[q] = s0 + r;
      }
    }
    __syncthreads();  // (fill[] aliases xch: every scatter counter has been consumed before the sorted keys overwrite it)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (dst[q] >= 0) S.xch[dst[q]] = mine[q];
  } else if (nsel <= 1024) {
    int n2 = 64;
    while (n2 < nsel) n2 <<= 1;
    const unsigned long long sorted = nf_sort_n(tid < nsel ? S.sel[tid] : 0ull, S.xch, n2);
    __syncthreads();
    S.xch[tid] = sorted;
  } else {
    unsigned long long k0 = 2 * tid < nsel ? S.sel[2 * tid] : 0ull, k1 = 2 * tid + 1 < nsel ? S.sel[2 * tid + 1] : 0ull;
    __syncthreads();
    nf_sort2048(k0, k1, S.xch);
    S.xch[2 * tid] = k0;
    S.xch[2 * tid + 1] = k1;
  }
  __syncthreads();
  // ---- candidate records in score order: key, class-offset box (ops.py:289), area -- the operands of torchvision's nms
  for (int e = tid; e < nsel; e += 1024) {
    const unsigned long long key = S.xch[e];
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

// v_min_f32 / v_max_f32 as the hardware does them (IEEE mode: a NaN operand yields the other operand, like fminf / fmaxf), without the
// canonicalising `v_max x, x` the compiler puts in front of every fminf / fmaxf on values it cannot prove free of signalling NaNs.
__device__ __forceinline__ float nf_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b)); return r; }
__device__ __forceinline__ float nf_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b)); return r; }
__device__ __forceinline__ float nf_max0(float b) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(b)); return r; }
__device__ __forceinline__ float nf_readlane(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }

// One 256-thread workgroup per 64 x 64 tile (cb, rb), its 4 waves take 16 rows each; workgroup g of an image walks the tiles
// g, g + gridDim.x, ... of the image's OWN triangle (nothing is launched per tile, so an image with few candidates costs a few tiles).
// Row operands are wave-uniform (v_readlane of the row box held by lane i), so every test is arithmetic against scalar operands, and
// it is torchvision's expression itself -- inter = max(0, w) * max(0, h), union = (area_i + area_j) - inter -- with the division
// replaced by q = inter * rcp(union): |q - thr| > band decides (yes = q > thr); only when some lane of the wave lands inside the
// band (or q is NaN: 0/0, NaN boxes), or an area is not a sane positive number, the division itself runs for that row.
// Why that is exact: with both areas > 1e-30 the union is a normal number >= ~max(area) (inter <= min(area) up to rounding), so q is
// within ~2.5 ulp (3e-7 relative) of inter / union and the correctly rounded quotient within 0.5 ulp of it: outside
// band = 1e-6 * thr the comparison `fl(inter / union) > thr` is already decided.  (thr < 1e-6: band = inf, always the division.)
#define NF_MASK_WG 64
__global__ __launch_bounds__(256) void nf_mask_kernel(float thr, float band, const float4* __restrict__ cbox_all, const float* __restrict__ area_all,
                                                      const NfMeta* __restrict__ meta_all, unsigned long long* __restrict__ mask_all, size_t img_bytes, int nbm) {
  __shared__ unsigned part[2][4][64];
  const size_t ioff = (size_t)blockIdx.y * img_bytes;
  const int nsel = reinterpret_cast<const NfMeta*>(reinterpret_cast<const char*>(meta_all) + ioff)->n_sel;
  const int NBall = (nsel + 63) >> 6, NB = NBall < nbm ? NBall : nbm, ntiles = NB * (NB + 1) / 2;  // only the first nbm column blocks get a bit matrix (see nf_resolve_kernel)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const float4* __restrict__ cbox = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(cbox_all) + ioff);
  const float* __restrict__ carea = reinterpret_cast<const float*>(reinterpret_cast<const char*>(area_all) + ioff);
  unsigned long long* __restrict__ mask = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(mask_all) + ioff);
  int par = 0;
  for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x, par ^= 1) {
    int cb = (int)((__builtin_sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);  // t = cb (cb + 1) / 2 + rb, 0 <= rb <= cb
    while (cb * (cb + 1) / 2 > t) --cb;
    while ((cb + 1) * (cb + 2) / 2 <= t) ++cb;
    const int rb = t - cb * (cb + 1) / 2;
    const int j = cb * 64 + lane, i = rb * 64 + lane;
    const bool valid = j < nsel;
    const float4 me = valid ? cbox[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float marea = valid ? carea[j] : 1.f;
    // Rows / columns whose area is not a sane positive finite number (tiny, zero, negative, huge, NaN) make the whole tile take the
    // division (band = inf): with every area in (1e-30, 1e30) all extents are finite, the union is a normal number >= ~max(area) and q
    // is never NaN.  (Rows >= nsel -- only on the last diagonal tile -- hold stale records: their bits are masked below, whatever they are.)
    const float rar_l = i < nsel ? carea[i] : 1.f;
    const bool sane = __builtin_amdgcn_ballot_w64(!(rar_l > 1e-30f && rar_l < 1e30f) || !(marea > 1e-30f && marea < 1e30f)) == 0ull;
    const float band_t = sane ? band : INFINITY;
    // the wave's 16 row boxes: wave-uniform addresses -> wide scalar loads, the operands below are SGPRs (no per-row lane broadcasts)
    const float4* __restrict__ rp = cbox + (rb * 64 + 16 * wave);
    const float* __restrict__ ap = carea + (rb * 64 + 16 * wave);
    unsigned bits = 0u;  // bit (15 - r) = row r (rows are shifted in from the right; reversed below)
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 8) {
      float4 rbx[8];
      float rar[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) { rbx[g] = rp[r0 + g]; rar[g] = ap[r0 + g]; }
      float inter[8], uni[8], d[8];
      float closest = INFINITY;  // min over the 8 rows of |q - thr|
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float w = nf_max0(__fsub_rn(nf_min(rbx[g].z, me.z), nf_max(rbx[g].x, me.x)));
        const float h = nf_max0(__fsub_rn(nf_min(rbx[g].w, me.w), nf_max(rbx[g].y, me.y)));
        inter[g] = __fmul_rn(w, h);
        uni[g] = __fsub_rn(__fadd_rn(rar[g], marea), inter[g]);
        d[g] = __fsub_rn(__fmul_rn(inter[g], __builtin_amdgcn_rcpf(uni[g])), thr);
      }
#pragma unroll
      for (int g = 0; g < 8; g += 2) closest = __builtin_fminf(closest, __builtin_fminf(__builtin_fabsf(d[g]), __builtin_fabsf(d[g + 1])));  // (v_min3_f32)
      if (__builtin_amdgcn_ballot_w64(closest > band_t) != ~0ull) {  // (rare, wave-uniform) some lane is not sure: the reference's own expression
#pragma unroll
        for (int g = 0; g < 8; ++g) d[g] = __fdiv_rn(inter[g], uni[g]) > thr ? 1.f : -1.f;  // (inter, uni ARE the reference's operands)
      }
#pragma unroll
      for (int g = 0; g < 8; ++g)  // bits = 2 * bits + (d > 0): compare into VCC, add with carry-in
        asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(d[g]) : "vcc");
    }
    bits = __builtin_bitreverse32(bits) >> 16;  // row r -> bit r
    if (rb == cb) {  // the diagonal tile: only EARLIER candidates (rows 16 * wave + r < lane) count
      const int nlow = lane - 16 * wave;
      bits &= nlow >= 16 ? 0xFFFFu : nlow <= 0 ? 0u : ((1u << nlow) - 1u);
    }
    part[par][wave][lane] = bits;
    __syncthreads();  // (one barrier per tile: the partials alternate between two buffers)
    if (wave == 0) {
      const unsigned long long word = (unsigned long long)part[par][0][lane] | ((unsigned long long)part[par][1][lane] << 16) |
                                      ((unsigned long long)part[par][2][lane] << 32) | ((unsigned long long)part[par][3][lane] << 48);
      mask[(size_t)t * 64 + lane] = valid ? word : 0ull;
    }
  }
}

struct NfResTail {  // the on-the-fly part (column blocks beyond the bit matrix): shares the 32 KB of `buf`
  float4 kbox[1024];                 // class-offset boxes of the kept candidates so far, in kept order (max_det <= 1024 on the fast path)
  float karea[1024];
  float4 bbox[64];                   // the current block's candidates
  float barea[64];
  unsigned long long dg[16][64];     // per wave: 4 rows of the block's own 64 x 64 suppression tile
};
struct NfResShared {
  union {
    unsigned long long buf[8][8][64];  // the mask words among the 8 column blocks in flight: [c][rb - cb0] (32 KB)
    NfResTail tail;
  };
  unsigned long long hit[16][64];    // phase A: per wave, "an earlier kept box (of an earlier group of 8 blocks) overlaps me"
  unsigned long long kw[NF_NB];      // kept bits per column block
  int kpre[NF_NB + 1];               // kept boxes before each block
  int nkept, stop;
};

// ... and, for an image the K best candidates could not complete (fewer than max_det kept while more candidates exist), the general
// algorithm (nms_general_body) right here, from scratch, in the same workgroup: no extra launch on the common path.
__global__ __launch_bounds__(1024) void nf_resolve_kernel(int nc, int A, long P, long nkeys, const float* __restrict__ boxsrc, long img_stride, float iou_thres, int max_det,
                                                          int max_nms, float max_wh, int agnostic, int target, int cap, int partition,
                                                          const unsigned long long* __restrict__ gkeys, const int* __restrict__ cls_id, char* __restrict__ scratch,
                                                          size_t img_bytes, float* __restrict__ out_boxes, int* __restrict__ out_count, int* __restrict__ out_index, int nbm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(sizeof(NfResTail) <= sizeof(unsigned long long) * 8 * 8 * 64, "the tail state must fit the mask-word buffer");
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
  const int NBm = NB < nbm ? NB : nbm;  // column blocks covered by the bit matrix (nbm is a multiple of 8)
  for (int cb0 = 0; cb0 < NBm; cb0 += 8) {
    const int ncb = min(8, NBm - cb0);
    {
      // phase A (all 16 waves): the kept-words of every EARLIER group of 8 blocks are final -- waves (c, c + 8) fold the words
      // (cb0 + c, rb < cb0) against them straight from memory (independent loads), half of the row blocks each
      const int c = wave & 7, cb = cb0 + c;
      unsigned long long acc = 0ull;
      if (c < ncb) {
#pragma unroll 4
        for (int rb = wave >> 3; rb < cb0; rb += 2) acc |= mask[((size_t)(cb * (cb + 1) / 2 + rb)) * 64 + lane] & S.kw[rb];
      }
      S.hit[wave][lane] = acc;
      // ... and the words among the group itself go to LDS for the sequential part
      for (int q = wave; q < ncb * (ncb + 1) / 2; q += 16) {
        int cc = 0;
        while ((cc + 1) * (cc + 2) / 2 <= q) ++cc;
        const int r = q - cc * (cc + 1) / 2, cbq = cb0 + cc;
        S.buf[cc][r][lane] = mask[((size_t)(cbq * (cbq + 1) / 2 + cb0 + r)) * 64 + lane];
      }
    }
    __syncthreads();
    if (wave == 0) {  // the sequential part: one wave, no barrier inside (it reads kept-words it wrote itself)
      int nk = S.nkept;
      for (int c = 0; c < ncb && nk < max_det; ++c) {
        const int cb = cb0 + c;
        unsigned long long hit = S.hit[c][lane] | S.hit[c + 8][lane];
        for (int r = 0; r < c; ++r) hit |= S.buf[c][r][lane] & S.kw[cb0 + r];
        const bool alive = (cb * 64 + lane < nsel) && hit == 0ull;
        unsigned long long km = nms_resolve(S.buf[c][c][lane], alive);
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
  // ---- column blocks beyond the bit matrix (only when the first nbm * 64 candidates did not yield max_det boxes): a candidate is tested
  // against the KEPT boxes only (<= max_det of them, not against every earlier candidate) and against its own block, on the fly, with the
  // reference's expression itself (the decision the bit matrix encodes: fl(inter / union) > thr) -- a few thousand divisions per block
  if (!S.stop && NB > NBm) {  // (workgroup-uniform)
    NfResTail& T = S.tail;
    const float4* cbox = nf_cbox(img);
    const float* carea = nf_area(img);
    if (tid == 0) {
      int run = 0;
      for (int cb = 0; cb < NBm; ++cb) { S.kpre[cb] = run; run += __popcll(S.kw[cb]); }
    }
    __syncthreads();
    for (int e = tid; e < NBm * 64; e += 1024) {
      const unsigned long long w = S.kw[e >> 6];
      if ((w >> (e & 63)) & 1ull) {
        const int rank = S.kpre[e >> 6] + __popcll(w & ((1ull << (e & 63)) - 1ull));
        T.kbox[rank] = cbox[e];
        T.karea[rank] = carea[e];
      }
    }
    auto hits = [&](const float4& a, float aa, const float4& bx, float ab) {
      const float w = fmaxf(0.f, __fsub_rn(fminf(a.z, bx.z), fmaxf(a.x, bx.x)));
      const float h = fmaxf(0.f, __fsub_rn(fminf(a.w, bx.w), fmaxf(a.y, bx.y)));
      const float inter = __fmul_rn(w, h), uni = __fsub_rn(__fadd_rn(aa, ab), inter);
      return __fdiv_rn(inter, uni) > iou_thres;
    };
    for (int cb = NBm; cb < NB; ++cb) {
      const int j = cb * 64 + lane;
      const bool valid = j < nsel;
      const float4 me = valid ? cbox[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float marea = valid ? carea[j] : 1.f;
      if (wave == 0) { T.bbox[lane] = me; T.barea[lane] = marea; }
      __syncthreads();  // kept list (first pass: gathered above; later: appended by wave 0) and this block's boxes are visible
      const int nk = S.nkept;
      bool hit = false;
      for (int k = wave; k < nk; k += 16) hit = hit || hits(T.kbox[k], T.karea[k], me, marea);
      S.hit[wave][lane] = hit ? 1ull : 0ull;
      unsigned long long rows = 0ull;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * wave + q;
        if (r < lane && hits(T.bbox[r], T.barea[r], me, marea)) rows |= 1ull << r;  // only EARLIER candidates of the block count
      }
      T.dg[wave][lane] = rows;
      __syncthreads();
      if (wave == 0) {
        unsigned long long h2 = 0ull, word = 0ull;
#pragma unroll
        for (int w2 = 0; w2 < 16; ++w2) { h2 |= S.hit[w2][lane]; word |= T.dg[w2][lane]; }
        const bool alive = valid && h2 == 0ull;
        unsigned long long km = nms_resolve(word, alive);
        const int room = max_det - nk;
        if (__popcll(km) > room) {
          unsigned long long t = km;
          for (int r = 0; r < room; ++r) t &= t - 1ull;
          km ^= t;
        }
        if ((km >> lane) & 1ull) {  // append to the kept list
          const int pos = nk + __popcll(km & ((1ull << lane) - 1ull));
          T.kbox[pos] = me;
          T.karea[pos] = marea;
        }
        if (lane == 0) {
          S.kw[cb] = km;
          S.nkept = nk + __popcll(km);
          S.stop = S.nkept >= max_det;
        }
      }
      __syncthreads();
      if (S.stop) break;
    }
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
  // complete when max_det boxes were found or every candidate the reference would look at (all of them, capped at max_nms) was among the K
  const bool done = nk >= max_det || nsel == ntotal || nsel >= max_nms;  // (workgroup-uniform)
  if (tid == 0) {
    out_count[b] = nk;
    meta->done = done ? 1 : 0;
  }
  if (!done) {
    __syncthreads();  // (LDS changes hands; the rows written above are rewritten below)
    nms_general_body(smem, nc, A, P, nkeys, 0, boxsrc, img_stride, iou_thres, max_det, max_nms, max_wh, agnostic, target, cap, partition, 0, gkeys, cls_id, out_boxes,
                     out_count, out_index);
  }
}
