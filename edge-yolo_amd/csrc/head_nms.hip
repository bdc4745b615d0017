// K9+K10 head decode (DGQP quality + DFL expectation + anchor decode + score modulation) and K11 batched NMS.
#include "common.h"
#include "tune.h"

// ============================================================================ head decode, one pyramid level
// thread = one anchor.  64 box logits -> 4 x softmax(16) -> {DFL expectation, top-4 + mean -> FC(20->hid, ReLU)
// -> FC(hid->1, sigmoid)} ; scores = sigmoid(cls) * clamp(q, 1e-6, 1-1e-6) ; boxes = xywh * stride.  All fp32.
// 128 anchors per workgroup.  The anchors' logits are first staged into LDS with coalesced 16-byte loads (rows padded
// to an odd number of 16-byte units, so the per-anchor row reads that follow are bank-conflict free); a thread then
// decodes one anchor entirely from LDS and writes its 4+nc outputs, coalesced across threads, to pred (B,4+nc,A).
#define HD_ANCH 128
#define HD_MAXL 4
// Up to HD_MAXL pyramid levels in ONE launch: the 20x20 and 40x40 levels are a few hundred short-lived workgroups each, far too
// few to fill the chip on their own; behind the 80x80 level's 1600 workgroups they cost nothing.  blk0[l] = first block of level l.
struct HdLevels {
  int nl;
  int H[HD_MAXL], W[HD_MAXL], boxCs[HD_MAXL], clsCs[HD_MAXL], a_off[HD_MAXL], blk0[HD_MAXL + 1];
  float stride[HD_MAXL];
  const void* box[HD_MAXL];
  const void* cls[HD_MAXL];
  const float* w1[HD_MAXL];
  const float* b1[HD_MAXL];
  const float* w2[HD_MAXL];
  const float* b2[HD_MAXL];
};
// Optional fused NMS candidate build (predict mode, single label): per anchor the best class (first maximal index, ops.py:274) and the
// key (score_bits << 32) | (0xFFFFFFFF - anchor) when score > conf (and the class passes the filter), else 0 -- exactly what
// nms_score_kernel derives from `pred`, from the same fp32 values, so the NMS sees identical candidates without the (B,4+nc,A) tensor
// ever being written or read.  box4 = rows 0-3 of pred (cx,cy,w,h), compact (B,4,A).
struct HdNms {
  unsigned long long* keys;  // [B][P] or null (off)
  int* cls_id;               // [B][P]
  float* box4;               // [B][4][A]
  const uint8_t* mask;       // [nc] or null
  float conf;
  int P;
  int xyxy;  // end2end heads: rows 0-3 of pred = x1,y1,x2,y2 (decode_bboxes with xywh=False, head.py:163-165) instead of cx,cy,w,h
};
template <typename T>
__global__ __launch_bounds__(HD_ANCH) void head_decode_kernel(int B, HdLevels lv, int nc, int hid, float* __restrict__ pred, int A, int boxLs, int clsLs, int vec, HdNms nm) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < HD_MAXL; ++i)
    if (i < lv.nl && (int)blockIdx.x >= lv.blk0[i]) l = i;
  const int H = lv.H[l], W = lv.W[l], boxCs = lv.boxCs[l], clsCs = lv.clsCs[l], a_off = lv.a_off[l];
  const float stride = lv.stride[l];
  const T* __restrict__ box = (const T*)lv.box[l];
  const T* __restrict__ cls = (const T*)lv.cls[l];
  const float* __restrict__ w1 = lv.w1[l];
  const float* __restrict__ b1 = lv.b1[l];
  const float* __restrict__ w2 = lv.w2[l];
  const float* __restrict__ b2 = lv.b2[l];
  const int blk = (int)blockIdx.x - lv.blk0[l];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* s_box = reinterpret_cast<T*>(smem);           // [HD_ANCH][boxLs]  (scalar path only: boxLs == 0 on the vector path)
  T* s_cls = s_box + HD_ANCH * boxLs;              // [HD_ANCH][clsLs]
  const int HW = H * W;
  const long total = (long)B * HW;
  const long idx0 = (long)blk * HD_ANCH;
  const int tid = threadIdx.x;
  const int nrow = (int)min((long)HD_ANCH, total - idx0);
  if (vec) {
    // (box logits: each thread reads its own anchor's 64 contiguous values straight from global, 8 x 16 B in flight)
    const int cv = nc >> 3;
    for (int v = tid; v < nrow * cv; v += HD_ANCH) {
      const int row = v / cv, c8 = (v - row * cv) << 3;
      Vec8<T> t;
      t.load(cls + (idx0 + row) * clsCs + c8);
      t.store(s_cls + row * clsLs + c8);
    }
  } else {
    for (int v = tid; v < nrow * 64; v += HD_ANCH) { const int row = v >> 6, c = v & 63; s_box[row * boxLs + c] = box[(idx0 + row) * boxCs + c]; }
    for (int v = tid; v < nrow * nc; v += HD_ANCH) { const int row = v / nc, c = v - row * nc; s_cls[row * clsLs + c] = cls[(idx0 + row) * clsCs + c]; }
  }
  __syncthreads();
  if (tid >= nrow) return;
  const long idx = idx0 + tid;
  const int b = (int)(idx / HW), a = (int)(idx - (long)b * HW);
  const int ay = a / W, ax = a - ay * W;
  const T* bp = vec ? box + idx * boxCs : s_box + tid * boxLs;
  float stat[20], dist[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float l[16];
    {
      Vec8<T> v0, v1;
      v0.load(bp + s * 16);
      v1.load(bp + s * 16 + 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) { l[i] = v0.get(i); l[8 + i] = v1.get(i); }
    }
    float mx = l[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, l[i]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { l[i] = __expf(l[i] - mx); sum += l[i]; }
    const float inv = 1.f / sum;
    float e = 0.f, t0 = -1.f, t1 = -1.f, t2 = -1.f, t3 = -1.f, psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float pr = l[i] * inv;
      e += pr * (float)i;
      psum += pr;
      // insert into the sorted top-4 (descending)
      float v = pr, u;
      u = fmaxf(t0, v); v = fminf(t0, v); t0 = u;
      u = fmaxf(t1, v); v = fminf(t1, v); t1 = u;
      u = fmaxf(t2, v); v = fminf(t2, v); t2 = u;
      t3 = fmaxf(t3, v);
    }
    dist[s] = e;
    stat[s * 5 + 0] = t0; stat[s * 5 + 1] = t1; stat[s * 5 + 2] = t2; stat[s * 5 + 3] = t3;
    stat[s * 5 + 4] = psum * (1.f / 16.f);
  }
  float q = 1.f;
  if (w1) {
    float o = b2[0];
    // 20 -> hid -> 1 on the VALU: packed fp32 FMAs (v_pk_fma_f32: two lanes of the dot product per instruction, even / odd inputs in the
    // two halves) -- this file is compiled without FMA contraction for the NMS arithmetic, which had turned every multiply-add of
    // this loop into a separate v_mul + v_add
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 st2[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) st2[i] = f32x2{stat[2 * i], stat[2 * i + 1]};
    // The weights are wave-uniform: read straight from global memory with uniform addresses they arrive by SCALAR loads (s_load_dwordx4 /
    // x8 into SGPRs, the FMAs take them as scalar operands) -- no LDS broadcast read per 4 multiply-adds (320 ds_read_b128 per wave
    // before, the kernel's bound).  Same operands in the same order: bit-identical.
    const float* __restrict__ gw1 = w1;
    const float* __restrict__ gb1 = b1;
    const float* __restrict__ gw2 = w2;
#pragma unroll 4
    for (int j = 0; j < hid; ++j) {
      const f32x4* wr = reinterpret_cast<const f32x4*>(gw1 + j * 20);
      f32x2 h2 = {gb1[j], 0.f};
#pragma unroll
      for (int i4 = 0; i4 < 5; ++i4) {
        const f32x4 w4 = wr[i4];
        h2 = __builtin_elementwise_fma(f32x2{w4[0], w4[1]}, st2[2 * i4], h2);
        h2 = __builtin_elementwise_fma(f32x2{w4[2], w4[3]}, st2[2 * i4 + 1], h2);
      }
      o = __builtin_fmaf(gw2[j], fmaxf(h2[0] + h2[1], 0.f), o);
    }
    q = fminf(fmaxf(ey_sigmoid(o), 1e-6f), 1.f - 1e-6f);
  }
  const float cx0 = ax + 0.5f, cy0 = ay + 0.5f;
  const float x1 = cx0 - dist[0], y1 = cy0 - dist[1], x2 = cx0 + dist[2], y2 = cy0 + dist[3];
  const float bcx = (x1 + x2) * 0.5f * stride, bcy = (y1 + y2) * 0.5f * stride, bw = (x2 - x1) * stride, bh = (y2 - y1) * stride;
  const T* cp = s_cls + tid * clsLs;
  float best = 0.f;
  int bi = 0;
  if (pred) {
    float* pp = pred + (long)b * (4 + nc) * A + a_off + a;
    if (nm.xyxy) { pp[0] = x1 * stride; pp[(long)A] = y1 * stride; pp[2L * A] = x2 * stride; pp[3L * A] = y2 * stride; }  // dist2bbox(xywh=False) * strides
    else { pp[0] = bcx; pp[(long)A] = bcy; pp[2L * A] = bw; pp[3L * A] = bh; }
    if (vec) {
      for (int c8 = 0; c8 < nc; c8 += 8) {
        Vec8<T> t;
        t.load(cp + c8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float sc = ey_sigmoid(t.get(i)) * q;
          pp[(long)(4 + c8 + i) * A] = sc;
          if (c8 + i == 0 || sc > best) { best = sc; bi = c8 + i; }
        }
      }
    } else {
      for (int c = 0; c < nc; ++c) {
        const float sc = ey_sigmoid(to_f(cp[c])) * q;
        pp[(long)(4 + c) * A] = sc;
        if (c == 0 || sc > best) { best = sc; bi = c; }
      }
    }
  } else if (vec) {
    for (int c8 = 0; c8 < nc; c8 += 8) {
      Vec8<T> t;
      t.load(cp + c8);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float sc = ey_sigmoid(t.get(i)) * q;
        if (c8 + i == 0 || sc > best) { best = sc; bi = c8 + i; }
      }
    }
  } else {
    for (int c = 0; c < nc; ++c) {
      const float sc = ey_sigmoid(to_f(cp[c])) * q;
      if (c == 0 || sc > best) { best = sc; bi = c; }
    }
  }
  if (nm.keys) {
    const int ga = a_off + a;
    float* bx = nm.box4 + (long)b * 4 * A + ga;
    bx[0] = bcx; bx[(long)A] = bcy; bx[2L * A] = bw; bx[3L * A] = bh;
    unsigned long long key = 0ull;
    if (best > nm.conf && (!nm.mask || nm.mask[bi]))
      key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)ga);
    nm.keys[(long)b * nm.P + ga] = key;
    nm.cls_id[(long)b * nm.P + ga] = bi;
  }
}

static int hd_pad(int elems, int es) {  // row stride (elements): 16-byte aligned, odd number of 16-byte units
  int units = (elems * es + 15) / 16;
  if (!(units & 1)) ++units;
  return units * 16 / es;
}

static int head_decode_impl(int dtype, int B, int nlevels, const int* H, const int* W, const float* stride, const void* const* box, const int* box_cstride,
                            const void* const* cls, const int* cls_cstride, int nc, const float* const* q_w1, const float* const* q_b1,
                            const float* const* q_w2, const float* const* q_b2, int q_hidden, float* pred, int A_total, const int* a_off, HdNms nm, ey_stream_t stream) {
  EY_CHECK((pred || nm.keys) && H && W && stride && box && cls && box_cstride && cls_cstride && a_off, "head_decode: null pointer");
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "head_decode: bad dtype");
  EY_CHECK(nlevels >= 1 && nlevels <= HD_MAXL, "head_decode: %d levels (1..%d)", nlevels, HD_MAXL);
  EY_CHECK(B > 0 && nc > 0, "head_decode: bad extent");
  const int es = dtype == EY_F16 ? 2 : 4;
  const bool quality = q_w1 && q_w1[0];
  HdLevels lv;
  lv.nl = nlevels;
  int vec = nc % 8 == 0;
  long nblk = 0;
  for (int l = 0; l < HD_MAXL; ++l) {
    const int s = l < nlevels ? l : 0;
    EY_CHECK(box[s] && cls[s] && H[s] > 0 && W[s] > 0, "head_decode: level %d: bad extent / null pointer", s);
    EY_CHECK(box_cstride[s] >= 64 && cls_cstride[s] >= nc, "head_decode: cstride");
    EY_CHECK(a_off[s] >= 0 && a_off[s] + H[s] * W[s] <= A_total, "head_decode: level [%d,%d) outside A=%d", a_off[s], a_off[s] + H[s] * W[s], A_total);
    EY_CHECK(!quality || (q_w1[s] && q_b1 && q_b1[s] && q_w2 && q_w2[s] && q_b2 && q_b2[s] && q_hidden > 0), "head_decode: incomplete quality head");
    lv.H[l] = H[s]; lv.W[l] = W[s]; lv.stride[l] = stride[s]; lv.box[l] = box[s]; lv.cls[l] = cls[s]; lv.boxCs[l] = box_cstride[s]; lv.clsCs[l] = cls_cstride[s];
    lv.a_off[l] = a_off[s];
    lv.w1[l] = quality ? q_w1[s] : nullptr; lv.b1[l] = quality ? q_b1[s] : nullptr; lv.w2[l] = quality ? q_w2[s] : nullptr; lv.b2[l] = quality ? q_b2[s] : nullptr;
    vec = vec && (box_cstride[s] * es) % 16 == 0 && (cls_cstride[s] * es) % 16 == 0 && ey_aligned(box[s], 16) && ey_aligned(cls[s], 16);
    lv.blk0[l] = (int)nblk;
    if (l < nlevels) nblk += ((long)B * H[s] * W[s] + HD_ANCH - 1) / HD_ANCH;
  }
  lv.blk0[HD_MAXL] = (int)nblk;
  EY_CHECK(nblk < (1L << 31), "head_decode: too many anchors");
  const int boxLs = vec ? 0 : hd_pad(64, es), clsLs = hd_pad(nc, es);
  const size_t lds = (size_t)HD_ANCH * (boxLs + clsLs) * es;  // (the quality-head weights are scalar operands: no LDS copy)
  EY_CHECK(lds <= 160 * 1024, "head_decode: nc=%d needs %zu B of LDS", nc, lds);
  dim3 grid((unsigned)nblk);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EY_F16) {
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)head_decode_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "head_decode: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(head_decode_kernel<f16>, grid, dim3(HD_ANCH), lds, st, B, lv, nc, quality ? q_hidden : 0, pred, A_total, boxLs, clsLs, vec, nm);
  } else {
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)head_decode_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "head_decode: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(head_decode_kernel<float>, grid, dim3(HD_ANCH), lds, st, B, lv, nc, quality ? q_hidden : 0, pred, A_total, boxLs, clsLs, vec, nm);
  }
  EY_LAUNCH_CHECK("ey_head_decode");
  return EY_OK;
}

extern "C" int ey_head_decode_levels(int dtype, int B, int nlevels, const int* H, const int* W, const float* stride, const void* const* box, const int* box_cstride,
                                     const void* const* cls, const int* cls_cstride, int nc, const float* const* q_w1, const float* const* q_b1,
                                     const float* const* q_w2, const float* const* q_b2, int q_hidden, float* pred, int A_total, const int* a_off, ey_stream_t stream) {
  EY_CHECK(pred, "head_decode: null pred");
  HdNms nm = {nullptr, nullptr, nullptr, nullptr, 0.f, 0, 0};
  return head_decode_impl(dtype, B, nlevels, H, W, stride, box, box_cstride, cls, cls_cstride, nc, q_w1, q_b1, q_w2, q_b2, q_hidden, pred, A_total, a_off, nm, stream);
}

extern "C" int ey_head_decode_levels_xyxy(int dtype, int B, int nlevels, const int* H, const int* W, const float* stride, const void* const* box, const int* box_cstride,
                                          const void* const* cls, const int* cls_cstride, int nc, const float* const* q_w1, const float* const* q_b1,
                                          const float* const* q_w2, const float* const* q_b2, int q_hidden, float* pred, int A_total, const int* a_off, ey_stream_t stream) {
  EY_CHECK(pred, "head_decode: null pred");
  HdNms nm = {nullptr, nullptr, nullptr, nullptr, 0.f, 0, 1};
  return head_decode_impl(dtype, B, nlevels, H, W, stride, box, box_cstride, cls, cls_cstride, nc, q_w1, q_b1, q_w2, q_b2, q_hidden, pred, A_total, a_off, nm, stream);
}

static int nms_pow2(int A);
static size_t nf_scratch_bytes(int B);
static size_t nms_cand_bytes(int B, int A) { return ((size_t)B * nms_pow2(A) * (8 + 4) + (size_t)B * 4 * A * 4 + 255) & ~(size_t)255; }  // keys + class ids + (cx,cy,w,h)
extern "C" size_t ey_nms_candidates_bytes(int B, int A) { return nms_cand_bytes(B, A) + nf_scratch_bytes(B); }  // + the scratch of the NMS fast path

extern "C" int ey_head_decode_levels_nms(int dtype, int B, int nlevels, const int* H, const int* W, const float* stride, const void* const* box,
                                         const int* box_cstride, const void* const* cls, const int* cls_cstride, int nc, const float* const* q_w1,
                                         const float* const* q_b1, const float* const* q_w2, const float* const* q_b2, int q_hidden, float* pred_or_null,
                                         int A_total, const int* a_off, float conf_thres, const uint8_t* class_mask, void* candidates, size_t candidates_bytes,
                                         ey_stream_t stream) {
  EY_CHECK(candidates && ey_aligned(candidates, 16) && candidates_bytes >= ey_nms_candidates_bytes(B, A_total), "head_decode_nms: candidate buffer missing / too small");
  EY_CHECK(conf_thres >= 0.f && conf_thres <= 1.f, "head_decode_nms: Invalid Confidence threshold %f, valid values are between 0.0 and 1.0", conf_thres);
  EY_CHECK(a_off && H && W, "head_decode_nms: null pointer");
  long covered = 0;
  for (int l = 0; l < nlevels; ++l) covered += (long)H[l] * W[l];
  EY_CHECK(covered == A_total, "head_decode_nms: the levels cover %ld of the %d anchors (every key slot must be written)", covered, A_total);
  const int P = nms_pow2(A_total);
  HdNms nm;
  nm.keys = (unsigned long long*)candidates;
  nm.cls_id = (int*)(nm.keys + (size_t)B * P);
  nm.box4 = (float*)(nm.cls_id + (size_t)B * P);
  nm.mask = class_mask;
  nm.conf = conf_thres;
  nm.P = P;
  nm.xyxy = 0;
  return head_decode_impl(dtype, B, nlevels, H, W, stride, box, box_cstride, cls, cls_cstride, nc, q_w1, q_b1, q_w2, q_b2, q_hidden, pred_or_null, A_total, a_off, nm, stream);
}

extern "C" int ey_head_decode(int dtype, int B, int H, int W, int nc, float stride, const void* box, int box_cstride, const void* cls,
                              int cls_cstride, const float* q_w1, const float* q_b1, const float* q_w2, const float* q_b2, int q_hidden,
                              float* pred, int A_total, int a_off, ey_stream_t stream) {
  EY_CHECK(box && cls && pred, "head_decode: null pointer");
  EY_CHECK(!q_w1 || (q_b1 && q_w2 && q_b2 && q_hidden > 0), "head_decode: incomplete quality head");
  return ey_head_decode_levels(dtype, B, 1, &H, &W, &stride, &box, &box_cstride, &cls, &cls_cstride, nc, &q_w1, &q_b1, &q_w2, &q_b2, q_hidden, pred, A_total, &a_off, stream);
}

// ============================================================================ NMS
// Stage 1 (nms_score): per anchor best class (first maximal index, ops.py:274) and confidence; anchors that pass
//   conf > thr (and the class filter) get the 64-bit key  (score_bits << 32) | (0xFFFFFFFF - anchor); others 0.
//   Sorting keys DESCENDING = scores descending with ties broken by ascending candidate (= anchor) order, i.e. the
//   stable descending sort torchvision.ops.nms applies to the compacted candidate list (ops.py:253,275,296).
// Stage 2 (nms_select_greedy_kernel, below): per image, radix-select the next chunk of best candidates, sort only that
//   chunk in LDS, run the greedy suppression on it with all 16 waves; repeat until max_det boxes are kept.
//   IoU arithmetic is the torchvision CPU kernel's, op by op, with explicit round-to-nearest intrinsics (no FMA
//   contraction), so decisions are bit-identical with the fp32 CPU oracle.
__global__ __launch_bounds__(256) void nms_score_kernel(int nc, int A, const float* __restrict__ pred, float conf_thres,
                                                        const uint8_t* __restrict__ class_mask, unsigned long long* __restrict__ keys, int* __restrict__ cls_id, int P) {
  const int b = blockIdx.y;
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= P) return;
  unsigned long long key = 0ull;
  if (a < A) {
    const float* pp = pred + (long)b * (4 + nc) * A + 4L * A + a;
    float best = pp[0];
    int bi = 0;
    for (int c = 1; c < nc; ++c) {
      const float v = pp[(long)c * A];
      if (v > best) { best = v; bi = c; }
    }
    if (best > conf_thres && (!class_mask || class_mask[bi]))
      key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)a);
    cls_id[(long)b * P + a] = bi;
  }
  keys[(long)b * P + a] = key;
}

// 4 anchors per thread with 16-byte loads (A % 4 == 0, pred 16-byte aligned): a quarter of the load instructions of the kernel above;
// same comparisons in the same class order, so keys and class ids are identical.
__global__ __launch_bounds__(256) void nms_score4_kernel(int nc, int A, const float* __restrict__ pred, float conf_thres, const uint8_t* __restrict__ class_mask,
                                                         unsigned long long* __restrict__ keys, int* __restrict__ cls_id, int P) {
  const int b = blockIdx.y;
  const int a0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (a0 >= P) return;
  unsigned long long key[4] = {0ull, 0ull, 0ull, 0ull};
  if (a0 < A) {  // A % 4 == 0: the whole quad is inside
    const float* pp = pred + (long)b * (4 + nc) * A + 4L * A + a0;
    f32x4 best = *reinterpret_cast<const f32x4*>(pp);
    int bi[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (int c = 1; c < nc; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(pp + (long)c * A);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (v[j] > best[j]) { best[j] = v[j]; bi[j] = c; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (best[j] > conf_thres && (!class_mask || class_mask[bi[j]]))
        key[j] = ((unsigned long long)__float_as_uint(best[j]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(a0 + j));
    *reinterpret_cast<int4*>(cls_id + (long)b * P + a0) = make_int4(bi[0], bi[1], bi[2], bi[3]);
  }
  unsigned long long* kp = keys + (long)b * P + a0;
  *reinterpret_cast<ulonglong2*>(kp) = make_ulonglong2(key[0], key[1]);
  *reinterpret_cast<ulonglong2*>(kp + 2) = make_ulonglong2(key[2], key[3]);
}

// multi_label (validation-mode) keys: one candidate per (anchor, class) pair with score > conf, enumerated anchor-major
// like `torch.where(cls > conf_thres)` (ops.py:270-272): key = (score_bits << 32) | (0xFFFFFFFF - (a*nc + c)).
__global__ __launch_bounds__(256) void nms_score_ml_kernel(int nc, int A, const float* __restrict__ pred, float conf_thres,
                                                           const uint8_t* __restrict__ class_mask, unsigned long long* __restrict__ keys, long P) {
  const int b = blockIdx.y;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // = c*Apad + a  (a fastest: coalesced reads)
  const int Apad = (A + 255) / 256 * 256;
  const int c = (int)(idx / Apad), a = (int)(idx - (long)c * Apad);
  if (c >= nc || a >= A) return;
  const float v = pred[(long)b * (4 + nc) * A + (long)(4 + c) * A + a];
  unsigned long long key = 0ull;
  if (v > conf_thres && (!class_mask || class_mask[c]))
    key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(a * nc + c));
  keys[(long)b * P + (long)a * nc + c] = key;
}

struct KeptBox { float x1, y1, x2, y2, area; };

// torchvision CPU kernel arithmetic, op by op, round-to-nearest, no FMA contraction.  The early-out is exact: with
// w<=0 or h<=0 the reference computes inter=0 and 0/union (0, -0 or NaN), none of which exceeds thr in [0,1].
__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1, float bx2, float by2,
                                       float barea, float thr) {
  const float w = __fsub_rn(fminf(ax2, bx2), fmaxf(ax1, bx1));
  const float h = __fsub_rn(fminf(ay2, by2), fmaxf(ay1, by1));
  if (!(w > 0.f) || !(h > 0.f)) {
    if (w != w || h != h) {  // NaN coordinates: fall through to the reference's full expression
      const float ww = fmaxf(0.f, w), hh = fmaxf(0.f, h);
      const float inter = __fmul_rn(ww, hh);
      return __fdiv_rn(inter, __fsub_rn(__fadd_rn(aarea, barea), inter)) > thr;
    }
    return false;
  }
  const float inter = __fmul_rn(w, h);
  return __fdiv_rn(inter, __fsub_rn(__fadd_rn(aarea, barea), inter)) > thr;
}

// Stage 2: one 1024-thread workgroup per image.
//  (a) radix descent over the 64-bit keys from the top: a 4096-bin histogram of the current 12-bit digit picks the
//      next CHUNK of >= ~1024 best remaining candidates (bins are taken whole, highest first); a bin holding more than
//      NMS_CAP keys is opened on its next digit (keys are unique, so the descent terminates);
//  (b) the chunk is gathered into LDS and bitonic-sorted (descending key = descending score, ties by ascending anchor);
//  (c) greedy suppression over the sorted chunk, 1024 candidates at a time: every thread holds one candidate and
//      first tests it against the kept list (parallel), then the 16 waves resolve their 64 candidates in order,
//      each kept box being published to LDS and tested by all later candidates.
//  Stops at max_det kept, max_nms candidates, or when the keys are exhausted.  Cost follows the number of
//  candidates actually needed, not A: no full sort.
#define NMS_CAP 4096
#define NMS_TARGET 2048
#define NMS_MODE_XYXY 1  // box rows are x1,y1,x2,y2
#define NMS_MODE_TOPK 2  // keep the best max_det candidates in score order, no suppression

struct NmsShared {
  unsigned hist[4096];
  unsigned long long chunk[NMS_CAP];
  KeptBox wbox[16][64];            // per wave: its 64 candidates' boxes (broadcast reads in the pairwise pass)
  unsigned long long wsup[16][64];  // per wave: bit i of wsup[j] = candidate i (earlier, alive) suppresses candidate j
  int sel_lo, cnt, nkept, processed;
  // class-partitioned greedy (per-class NMS: candidates of different classes never interact, so wave w resolves the classes
  // c % 16 == w on its own, with no workgroup barrier inside the suppression)
  unsigned short order[1024];   // candidate indices of the chunk grouped by owner wave, score order inside a group
  unsigned char keptflag[1024];
  int wcnt[16][16];             // [wave][owner]: candidates of that owner in that wave's 64 (then: exclusive prefix over the waves)
  int ototal[16], obase[16], npk[16], wtot2[16];
  float wlo[16], whi[16];
  int wbad[16];
  float rmin, rmax;             // x extent of every candidate seen so far
  int part;                     // 1 while the partitioned path is valid for this image
  int part_chunk;               // ... and worth it for this chunk (no owner holds more than two 64-candidate rounds)
};

__device__ __forceinline__ void nms_block_suffix_scan(unsigned* h) {
  // in place: h[b] <- sum_{b' >= b} h[b'] for 4096 bins, 1024 threads (4 bins each): per-thread suffix, wave-level suffix scan of the
  // thread totals by lane shuffles (no barrier), then the 16 wave totals through LDS -- 2 barriers in all
  __shared__ unsigned wtot[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned v3 = h[4 * t + 3], v2 = h[4 * t + 2] + v3, v1 = h[4 * t + 1] + v2, v0 = h[4 * t] + v1;
  unsigned run = v0;  // inclusive suffix sum over lanes >= lane
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_down(run, off, 64);
    if (lane + off < 64) run += o;
  }
  if (lane == 0) wtot[wave] = run;
  __syncthreads();
  unsigned above = run - v0;  // lanes above mine in this wave
  for (int w = wave + 1; w < 16; ++w) above += wtot[w];
  h[4 * t] = v0 + above; h[4 * t + 1] = v1 + above; h[4 * t + 2] = v2 + above; h[4 * t + 3] = v3 + above;
  __syncthreads();
}

// Descending bitonic sort of 1024 keys, one per thread: the 45 compare-exchange steps with partner distance < 64 run on lane shuffles
// inside the wave (no barrier, no LDS), only the 10 steps with distance >= 64 go through LDS.  Keys are unique (or 0 = padding), so
// the result does not depend on the network.
__device__ __forceinline__ unsigned long long nms_sort1024(unsigned long long key, unsigned long long* xch) {
  const int t = threadIdx.x;
#pragma unroll 1
  for (int k = 2; k <= 1024; k <<= 1) {
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

// "keep candidate j iff it is alive and no KEPT earlier candidate of the wave suppresses it" (mysup: bit i = earlier candidate i overlaps
// me).  The greedy answer is the unique fixed point of  K[j] = alive[j] & ((mysup[j] & K) == 0)  -- by induction over j any fixed point
// equals the sequential result -- and iterating from K = alive fixes one more leading position per round at worst, in practice the depth
// of the suppression chains (a handful): a few wave-wide ballots instead of a 64-step scalar scan.  Terminates when K stops changing.
__device__ __forceinline__ unsigned long long nms_resolve(unsigned long long mysup, bool alive) {
  unsigned long long K = __ballot(alive);
  for (int it = 0; it < 64; ++it) {
    const unsigned long long Kn = __ballot(alive && (mysup & K) == 0ull);
    if (Kn == K) break;
    K = Kn;
  }
  return K;
}

__device__ __forceinline__ void nms_general_body(char* smem, int nc, int A, long P, long nkeys, int multi_label, const float* __restrict__ boxsrc, long img_stride,
                                                 float iou_thres, int max_det, int max_nms, float max_wh, int agnostic, int target, int cap, int partition, int mode,
                                                 const unsigned long long* __restrict__ gkeys, const int* __restrict__ cls_id,
                                                 float* __restrict__ out_boxes, int* __restrict__ out_count,
                                                 int* __restrict__ out_index) {
  NmsShared& S = *reinterpret_cast<NmsShared*>(smem);
  KeptBox* kept = reinterpret_cast<KeptBox*>(smem + sizeof(NmsShared));
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long* keys = gkeys + (long)b * P;
  const float* pb = boxsrc + (long)b * img_stride;  // rows cx | cy | w | h, A values each (pred, or the compact box4 of the fused decode)
  const int* cid = cls_id + (long)b * P;  // (single-label only)

  // radix-descent state (wave-uniform, identical in every thread)
  const int SHIFT[6] = {51, 39, 27, 15, 3, 0};
  const int BITS[6] = {12, 12, 12, 12, 12, 3};
  int level = 0, hi[6];
  unsigned long long prefix[6];
  hi[0] = 4096;
  prefix[0] = 0ull;
  if (tid == 0) { S.nkept = 0; S.processed = 0; S.part = partition; S.rmin = INFINITY; S.rmax = -INFINITY; }
  if (tid < 16) S.npk[tid] = 0;
  unsigned short* pkl = reinterpret_cast<unsigned short*>(kept + max_det);  // [16][max_det]: ranks of the boxes each owner wave has kept so far
  // result rows are collected in LDS and written once at the end: a global store inside the suppression loop would sit on the critical
  // path of every round (the next workgroup barrier waits for it to complete)
  float* orow = reinterpret_cast<float*>(pkl + (partition ? 16 * max_det : 0));  // [max_det][6]
  int* oidx = reinterpret_cast<int*>(orow + 6 * max_det);                          // [max_det]
  KeptBox* cbox = &S.wbox[0][0];                                             // [1024]: the chunk's candidate boxes (class offset applied)
  bool need_hist = true;
  __syncthreads();

  while (true) {
    if (S.nkept >= max_det || S.processed >= max_nms) break;
    const int sh = SHIFT[level], nb = 1 << BITS[level];
    const unsigned long long above_mask = level == 0 ? 0ull : (~0ull << SHIFT[level - 1]);
    if (need_hist) {
      for (int i = tid; i < 4096; i += 1024) S.hist[i] = 0u;
      __syncthreads();
      for (long i = tid; i < nkeys; i += 1024) {
        const unsigned long long k = keys[i];
        if (k != 0ull && (k & above_mask) == prefix[level]) atomicAdd(&S.hist[(unsigned)(k >> sh) & (nb - 1)], 1u);
      }
      __syncthreads();
      nms_block_suffix_scan(S.hist);  // hist[b] = #keys (under this prefix) with digit >= b
      need_hist = false;
    }
    const int H = hi[level];
    const unsigned s_hi = H < 4096 ? S.hist[H] : 0u;  // keys already consumed at this level (digit >= H)
    const unsigned remaining = S.hist[0] - s_hi;
    if (H == 0 || remaining == 0u) {  // this subtree is exhausted
      if (level == 0) break;
      --level;
      need_hist = true;
      __syncthreads();  // every wave has read S.hist[H] / S.hist[0] before the next iteration zeroes the histogram
      continue;
    }
    // lowest-cost chunk: the largest bin_lo < H whose digit range [bin_lo, H) holds >= NMS_TARGET keys (else 0)
    if (tid == 0) S.sel_lo = 0;
    __syncthreads();
    {
      int best = 0;  // bins are visited in increasing order: the last qualifying one is this thread's largest
      for (int bin = tid; bin < H; bin += 1024)
        if (S.hist[bin] - s_hi >= (unsigned)target) best = bin;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off, 64));
      if (lane == 0 && best > 0) atomicMax(&S.sel_lo, best);  // one LDS atomic per wave
    }
    __syncthreads();
    int lo = S.sel_lo;
    unsigned cnt = S.hist[lo] - s_hi;
    if (cnt > (unsigned)cap) {
      ++lo;  // drop the heavy lowest bin: what is left has < target keys
      cnt = lo < H ? S.hist[lo] - s_hi : 0u;
      if (cnt == 0u) {  // the top remaining bin alone exceeds the cap: open it on its next digit
        const int heavy = lo - 1;
        hi[level] = heavy;  // when we come back, continue below it
        prefix[level + 1] = prefix[level] | ((unsigned long long)heavy << sh);
        ++level;
        hi[level] = 1 << BITS[level];
        need_hist = true;
        __syncthreads();
        continue;
      }
    }
    hi[level] = lo;
    // ---- gather the chunk (unordered), then sort it
    if (tid == 0) S.cnt = 0;
    __syncthreads();
    for (long i = tid; i < nkeys; i += 1024) {
      const unsigned long long k = keys[i];
      if (k != 0ull && (k & above_mask) == prefix[level]) {
        const int dgt = (int)((unsigned)(k >> sh) & (nb - 1));
        if (dgt >= lo && dgt < H) S.chunk[atomicAdd(&S.cnt, 1)] = k;
      }
    }
    __syncthreads();
    int n = (int)cnt;
    if (n <= 1024) {  // (predict mode: cap = 1024) one key per thread, register sort
      const unsigned long long sorted = nms_sort1024(tid < n ? S.chunk[tid] : 0ull, S.chunk + 1024);
      __syncthreads();
      S.chunk[tid] = sorted;
      __syncthreads();
    } else {
      int n2 = 64;
      while (n2 < n) n2 <<= 1;
      for (int i = n + tid; i < n2; i += 1024) S.chunk[i] = 0ull;
      __syncthreads();
      for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int t = tid; t < (n2 >> 1); t += 1024) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;
            const unsigned long long u = S.chunk[i], v = S.chunk[ixj];
            const bool desc = (i & k) == 0;
            if (desc ? (u < v) : (u > v)) { S.chunk[i] = v; S.chunk[ixj] = u; }
          }
          __syncthreads();
        }
      }
    }
    if (S.processed + n > max_nms) n = max_nms - S.processed;  // ops.py:285-286 cap
    __syncthreads();
    if (tid == 0) S.processed += n;

    // ---- greedy suppression over the sorted chunk
    for (int sb = 0; sb < n; sb += 1024) {
      __syncthreads();
      if (S.nkept >= max_det) break;
      const int i = sb + tid;
      bool alive = i < n;
      float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, area = 0.f, conf = 0.f, ux1 = 0.f, uy1 = 0.f, ux2 = 0.f, uy2 = 0.f;
      int a = 0, ci = 0;
      auto load_cand = [&](unsigned long long key) {
        a = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        conf = __uint_as_float((unsigned)(key >> 32));
        if (multi_label) { ci = a % nc; a = a / nc; }
        const float cx = pb[a], cy = pb[(long)A + a], w = pb[2L * A + a], h = pb[3L * A + a];
        if (mode & NMS_MODE_XYXY) { ux1 = cx; uy1 = cy; ux2 = w; uy2 = h; }  // rows already hold x1,y1,x2,y2 (end2end heads)
        else {
          const float hw = __fmul_rn(w, 0.5f), hh = __fmul_rn(h, 0.5f);  // xywh2xyxy, ops.py:430-432 (x/2 is exact)
          ux1 = __fsub_rn(cx, hw); uy1 = __fsub_rn(cy, hh); ux2 = __fadd_rn(cx, hw); uy2 = __fadd_rn(cy, hh);
        }
        if (!multi_label) ci = cid[a];
        const float off = agnostic ? 0.f : __fmul_rn((float)ci, max_wh);  // ops.py:289
        x1 = __fadd_rn(ux1, off); y1 = __fadd_rn(uy1, off); x2 = __fadd_rn(ux2, off); y2 = __fadd_rn(uy2, off);
        area = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
      };
      if (alive) load_cand(S.chunk[i]);
      if (mode & NMS_MODE_TOPK) {  // no suppression: the sorted candidates ARE the result rows (Detect.postprocess, head.py:167-189)
        const int nk0 = S.nkept, rank = nk0 + tid;
        if (alive && rank < max_det) {
          float* ob = orow + rank * 6;
          ob[0] = ux1; ob[1] = uy1; ob[2] = ux2; ob[3] = uy2; ob[4] = conf; ob[5] = (float)ci;
          oidx[rank] = a;
        }
        __syncthreads();
        if (tid == 0) S.nkept = min(max_det, nk0 + min(1024, n - sb));
        continue;
      }
      if (partition && n <= 1024 && S.part) {
        // ---- class-partitioned greedy.  Valid while all candidates seen so far span less than max_wh in x: boxes of different
        // classes (offset by c * max_wh, ops.py:289) then cannot overlap, exactly as in the reference.
        const unsigned long long lt = (1ull << lane) - 1ull;
        float lo = alive ? ux1 : INFINITY, hi = alive ? ux2 : -INFINITY;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off, 64)); hi = fmaxf(hi, __shfl_xor(hi, off, 64)); }
        const unsigned long long badm = __ballot(alive && !(ux1 == ux1 && ux2 == ux2));
        if (lane == 0) { S.wlo[wave] = lo; S.whi[wave] = hi; S.wbad[wave] = badm != 0ull; }
        cbox[tid] = KeptBox{x1, y1, x2, y2, area};
        S.keptflag[tid] = 0;
        const int owner = alive ? (ci & 15) : 16;
        int myrank = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
          const unsigned long long m = __ballot(owner == w);
          if (lane == 0) S.wcnt[wave][w] = __popcll(m);
          if (owner == w) myrank = __popcll(m & lt);
        }
        __syncthreads();
        if (tid < 16) {  // owner `tid`: exclusive prefix of its counts over the waves
          int run = 0;
          for (int wv = 0; wv < 16; ++wv) { const int c = S.wcnt[wv][tid]; S.wcnt[wv][tid] = run; run += c; }
          S.ototal[tid] = run;
        }
        if (tid == 32) {
          float mn = S.rmin, mx = S.rmax;
          int bad = 0;
          for (int wv = 0; wv < 16; ++wv) { mn = fminf(mn, S.wlo[wv]); mx = fmaxf(mx, S.whi[wv]); bad |= S.wbad[wv]; }
          S.rmin = mn; S.rmax = mx;
          if (bad || !(__fsub_rn(mx, mn) < max_wh)) S.part = 0;  // (also catches NaN / inf extents)
        }
        __syncthreads();
        if (tid == 0) {
          int run = 0, mx = 0;
          for (int w = 0; w < 16; ++w) { S.obase[w] = run; run += S.ototal[w]; mx = max(mx, S.ototal[w]); }
          // one dominant class would leave its wave to resolve everything alone (measured: 3x slower than the cooperative scan on the
          // random-init model, whose arg-max class is nearly constant): take the partitioned path only for balanced chunks
          S.part_chunk = S.part && mx <= 128;
        }
        __syncthreads();
        if (S.part_chunk) {
          if (alive) S.order[S.obase[owner] + S.wcnt[wave][owner] + myrank] = (unsigned short)tid;
          __syncthreads();
          {  // wave `wave` owns the classes c % 16 == wave: its candidates, in score order, 64 at a time; no workgroup barrier in here
            const int base = S.obase[wave], tot = S.ototal[wave], npk = S.npk[wave];
            const unsigned short* mypk = pkl + wave * max_det;
            int nck = 0;
            for (int t0 = 0; t0 < tot; t0 += 64) {
              const bool on = t0 + lane < tot;
              const int idx = on ? (int)S.order[base + t0 + lane] : 0;
              __builtin_amdgcn_wave_barrier();
              const KeptBox me = cbox[idx];
              bool al = on;
              for (int k = 0; k < npk; ++k) {  // boxes this owner kept in earlier chunks
                const KeptBox kb = kept[mypk[k]];
                al = al && !iou_gt(kb.x1, kb.y1, kb.x2, kb.y2, kb.area, me.x1, me.y1, me.x2, me.y2, me.area, iou_thres);
              }
              for (int k = 0; k < nck; ++k) {  // ... and earlier in this chunk (compacted to the front of its order[] segment)
                const KeptBox kb = cbox[S.order[base + k]];
                al = al && !iou_gt(kb.x1, kb.y1, kb.x2, kb.y2, kb.area, me.x1, me.y1, me.x2, me.y2, me.area, iou_thres);
              }
              unsigned long long mysup = 0ull;
              const int nin = min(64, tot - t0);
              for (int i = 0; i + 1 < nin; ++i) {  // pairwise inside the 64: bit i = "earlier candidate i overlaps me"
                const KeptBox bi = cbox[__builtin_amdgcn_readlane(idx, i)];
                if (i < lane && iou_gt(bi.x1, bi.y1, bi.x2, bi.y2, bi.area, me.x1, me.y1, me.x2, me.y2, me.area, iou_thres)) mysup |= 1ull << i;
              }
              const unsigned long long keptm = nms_resolve(mysup, al);
              if ((keptm >> lane) & 1ull) {
                S.order[base + nck + __popcll(keptm & lt)] = (unsigned short)idx;
                S.keptflag[idx] = 1;
              }
              nck += __popcll(keptm);
              __builtin_amdgcn_wave_barrier();
            }
          }
          __syncthreads();
          // kept candidates -> global ranks in score order (= chunk order), outputs, per-owner lists for the next chunk
          const int nk0 = S.nkept;
          const bool k_i = alive && S.keptflag[tid] != 0;
          const unsigned long long km = __ballot(k_i);
          if (lane == 0) S.wtot2[wave] = __popcll(km);
          __syncthreads();
          int before = nk0 + __popcll(km & lt), total = nk0;
          for (int w = 0; w < 16; ++w) { const int c = S.wtot2[w]; total += c; if (w < wave) before += c; }
          if (k_i && before < max_det) {
            kept[before] = KeptBox{x1, y1, x2, y2, area};
            float* ob = orow + before * 6;
            ob[0] = ux1; ob[1] = uy1; ob[2] = ux2; ob[3] = uy2; ob[4] = conf; ob[5] = (float)ci;
            oidx[before] = a;
            pkl[owner * max_det + atomicAdd(&S.npk[owner], 1)] = (unsigned short)before;
          }
          __syncthreads();
          if (tid == 0) S.nkept = min(max_det, total);
          continue;
        }
      }
      int checked = 0, nloc = min(1024, n - sb);
      const int nk0 = S.nkept;  // (uniform: the workgroup passed a barrier since the last write)
      if (n <= 1024 && nk0 >= 48 && nloc > 128) {
        // A later chunk: most candidates are already suppressed by boxes kept from earlier chunks (dense scenes keep ~1 in 4).  Test
        // everybody against those boxes once, in parallel, and COMPACT the survivors (order kept), so that the sequential rounds
        // below run over 64 live candidates each instead of 64 mostly dead ones.
        bool sup = false;
        if (alive) {
          int k = 0;
          for (; k + 4 <= nk0; k += 4) {
            const KeptBox k0 = kept[k], k1 = kept[k + 1], k2 = kept[k + 2], k3 = kept[k + 3];
            sup |= iou_gt(k0.x1, k0.y1, k0.x2, k0.y2, k0.area, x1, y1, x2, y2, area, iou_thres);
            sup |= iou_gt(k1.x1, k1.y1, k1.x2, k1.y2, k1.area, x1, y1, x2, y2, area, iou_thres);
            sup |= iou_gt(k2.x1, k2.y1, k2.x2, k2.y2, k2.area, x1, y1, x2, y2, area, iou_thres);
            sup |= iou_gt(k3.x1, k3.y1, k3.x2, k3.y2, k3.area, x1, y1, x2, y2, area, iou_thres);
          }
          for (; k < nk0; ++k) {
            const KeptBox kb = kept[k];
            sup |= iou_gt(kb.x1, kb.y1, kb.x2, kb.y2, kb.area, x1, y1, x2, y2, area, iou_thres);
          }
        }
        alive = alive && !sup;
        const unsigned long long am = __ballot(alive);
        if (lane == 0) S.wtot2[wave] = __popcll(am);
        __syncthreads();
        int before = __popcll(am & ((1ull << lane) - 1ull)), total = 0;
        for (int w = 0; w < 16; ++w) { const int c = S.wtot2[w]; total += c; if (w < wave) before += c; }
        unsigned long long* comp = S.chunk + 1024;  // (free in this mode: the chunk holds <= 1024 keys)
        if (alive) comp[before] = S.chunk[i];
        __syncthreads();
        nloc = total;
        alive = tid < nloc;
        if (alive) load_cand(comp[tid]);
        checked = nk0;
      }
      // pairwise pass, all 16 waves at once: bit i of mysup = "earlier candidate i of my wave overlaps me beyond the
      // threshold" (independent of who survives; survival is applied in the scan below)
      S.wbox[wave][lane] = KeptBox{x1, y1, x2, y2, area};
      __builtin_amdgcn_wave_barrier();
      unsigned long long mysup = 0ull;
      if (alive) {
#pragma unroll 4
        for (int i = 0; i < 63; ++i) {
          const KeptBox bi = S.wbox[wave][i];
          if (i < lane && iou_gt(bi.x1, bi.y1, bi.x2, bi.y2, bi.area, x1, y1, x2, y2, area, iou_thres)) mysup |= 1ull << i;
        }
      }
      const int nwaves = min(16, (nloc + 63) >> 6);
      for (int w = 0; w < nwaves; ++w) {
        __syncthreads();
        const int nk = S.nkept;
        if (nk >= max_det) break;
        if (wave >= w) {  // catch up with boxes kept since this candidate was last tested (4 at a time: no serial LDS chain)
          bool sup = false;
          int k = checked;
          for (; k + 4 <= nk; k += 4) {
            const KeptBox k0 = kept[k], k1 = kept[k + 1], k2 = kept[k + 2], k3 = kept[k + 3];
            sup |= iou_gt(k0.x1, k0.y1, k0.x2, k0.y2, k0.area, x1, y1, x2, y2, area, iou_thres);
            sup |= iou_gt(k1.x1, k1.y1, k1.x2, k1.y2, k1.area, x1, y1, x2, y2, area, iou_thres);
            sup |= iou_gt(k2.x1, k2.y1, k2.x2, k2.y2, k2.area, x1, y1, x2, y2, area, iou_thres);
            sup |= iou_gt(k3.x1, k3.y1, k3.x2, k3.y2, k3.area, x1, y1, x2, y2, area, iou_thres);
          }
          for (; k < nk; ++k) {
            const KeptBox kb = kept[k];
            sup |= iou_gt(kb.x1, kb.y1, kb.x2, kb.y2, kb.area, x1, y1, x2, y2, area, iou_thres);
          }
          alive = alive && !sup;
          checked = nk;
        }
        if (wave == w) {
          // Resolve this wave's 64 candidates: "keep j iff no KEPT earlier candidate suppresses it" is a scan over the
          // pairwise bit masks; the masks are pulled lane by lane with v_readlane (wave-uniform index), so the whole
          // dependent chain runs on the scalar unit.
          const unsigned long long keptm = nms_resolve(mysup, alive);
          const int rank = nk + __popcll(keptm & ((1ull << lane) - 1ull));
          const bool keep = ((keptm >> lane) & 1ull) && rank < max_det;
          if (keep) {
            kept[rank] = KeptBox{x1, y1, x2, y2, area};
            float* ob = orow + rank * 6;
            ob[0] = ux1; ob[1] = uy1; ob[2] = ux2; ob[3] = uy2; ob[4] = conf; ob[5] = (float)ci;
            oidx[rank] = a;
            if (partition) pkl[(ci & 15) * max_det + atomicAdd(&S.npk[ci & 15], 1)] = (unsigned short)rank;  // (a later chunk may run class-partitioned)
          }
          if (lane == 0) S.nkept = min(max_det, nk + __popcll(keptm));
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  const int nk = min(S.nkept, max_det);
  if (tid == 0) out_count[b] = nk;
  for (int r = tid; r < max_det * 6; r += 1024) out_boxes[(long)b * max_det * 6 + r] = r < nk * 6 ? orow[r] : 0.f;
  if (out_index)
    for (int r = tid; r < max_det; r += 1024) out_index[(long)b * max_det + r] = r < nk ? oidx[r] : -1;
}

__global__ __launch_bounds__(1024) void nms_select_greedy_kernel(int nc, int A, long P, long nkeys, int multi_label, const float* __restrict__ boxsrc, long img_stride,
                                                                 float iou_thres, int max_det, int max_nms, float max_wh, int agnostic, int target, int cap, int partition, int mode,
                                                                 const unsigned long long* __restrict__ gkeys, const int* __restrict__ cls_id,
                                                                 float* __restrict__ out_boxes, int* __restrict__ out_count,
                                                                 int* __restrict__ out_index) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  nms_general_body(smem, nc, A, P, nkeys, multi_label, boxsrc, img_stride, iou_thres, max_det, max_nms, max_wh, agnostic, target, cap, partition, mode, gkeys, cls_id, out_boxes,
                   out_count, out_index);
}

static int nms_pow2(int A) { return (A + 255) / 256 * 256; }  // key array length per image (padded for the score kernel grid)

#include "nms_fast.inc.h"
static size_t nf_scratch_bytes(int B) { return (size_t)B * nf_image_bytes(); }
static size_t nms_keys_bytes(int B, int A) { return ((size_t)B * nms_pow2(A) * (8 + 4) + 255) & ~(size_t)255; }  // keys + class ids

extern "C" size_t ey_nms_workspace_bytes(int B, int A) { return nms_keys_bytes(B, A) + nf_scratch_bytes(B); }  // + the fast path's scratch
extern "C" size_t ey_nms_workspace_bytes_ml(int B, int nc, int A) { return (size_t)B * (((size_t)A * nc + 255) / 256 * 256) * 8; }

// chunk size of the radix selection: [target, cap] candidates per round.  predict mode needs ~max_det survivors, so small chunks (one key
// per thread, register sort) are cheapest; validation mode (multi_label: up to max_nms = 30000 candidates) amortises each pass over the
// A*nc keys with the largest chunk the LDS sort takes.
static int nms_select_launch(int B, int nc, int A, long P, long nkeys, int multi_label, const float* boxsrc, long img_stride, float iou_thres, int max_det, int max_nms,
                             float max_wh, int agnostic, const unsigned long long* keys, const int* cls_id, float* out_boxes, int32_t* out_count, int32_t* out_index,
                             hipStream_t st, int mode = 0, char* fast_scratch = nullptr) {
  const int cap = (multi_label || max_det > 512) ? NMS_CAP : 1024, target = cap / 2;
  // per-class NMS with small chunks: the class-partitioned greedy (wave w resolves the classes c % 16 == w without workgroup barriers)
  const int partition = !agnostic && nc > 1 && cap == 1024 && !(mode & NMS_MODE_TOPK);
  const size_t lds_general = sizeof(NmsShared) + (size_t)max_det * sizeof(KeptBox) + (partition ? (size_t)16 * max_det * sizeof(unsigned short) : 0) +
                             (size_t)max_det * 7 * sizeof(float) + 16;
  EY_CHECK(lds_general + 4096 <= 160 * 1024, "nms: max_det=%d needs %zu B of LDS", max_det, lds_general);
  // ---- fast path (predict mode): select K best -> all-pairs bit matrix on the whole chip -> mask-arithmetic resolve; an image it cannot
  // complete (fewer than max_det kept among the K best while more candidates exist) is redone by the general algorithm inside the resolve kernel
  const long fast_k = tune().nms_fast_k;
  if (fast_scratch && fast_k > 0 && !multi_label && mode == 0 && max_det <= 1024 && max_nms >= NF_KMAX) {
    const int K = (int)(fast_k > NF_KMAX ? NF_KMAX : fast_k);
    const size_t ib = nf_image_bytes();
    if (nkeys <= 9 * 1024)  // 640x640: 8400 anchors -> the keys of an image stay in registers (9 per thread)
      hipLaunchKernelGGL(nf_select_kernel<9>, dim3(B), dim3(1024), 0, st, nc, A, P, nkeys, boxsrc, img_stride, max_nms, max_wh, agnostic, K, keys, cls_id, fast_scratch, ib);
    else
      hipLaunchKernelGGL(nf_select_kernel<0>, dim3(B), dim3(1024), 0, st, nc, A, P, nkeys, boxsrc, img_stride, max_nms, max_wh, agnostic, K, keys, cls_id, fast_scratch, ib);
    EY_LAUNCH_CHECK("ey_nms(select)");
    const float band = iou_thres >= 1e-6f ? (float)((double)iou_thres * 1e-6) : INFINITY;
    // bit matrix over the first nms_mask_k candidates only (a multiple of 512); the resolve kernel tests later ones against the kept boxes on the fly
    long mk = tune().nms_mask_k;
    if (mk <= 0 || mk > NF_KMAX) mk = NF_KMAX;
    const int nbm = (int)((mk + 511) / 512) * 8;
    const long mwg = tune().nms_mask_wg > 0 ? tune().nms_mask_wg : NF_MASK_WG;
    hipLaunchKernelGGL(nf_mask_kernel, dim3((unsigned)mwg, B), dim3(256), 0, st, iou_thres, band, (const float4*)(fast_scratch + (size_t)NF_KMAX * 8),
                       (const float*)(fast_scratch + (size_t)NF_KMAX * 24), (const NfMeta*)(fast_scratch + (size_t)NF_KMAX * 28 + (size_t)NF_TILES * 512),
                       (unsigned long long*)(fast_scratch + (size_t)NF_KMAX * 28), ib, nbm);
    EY_LAUNCH_CHECK("ey_nms(mask)");
    const size_t lds = lds_general > sizeof(NfResShared) ? lds_general : sizeof(NfResShared);
    if (lds > 60 * 1024 && hipFuncSetAttribute((const void*)nf_resolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "nms: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(nf_resolve_kernel, dim3(B), dim3(1024), lds, st, nc, A, P, nkeys, boxsrc, img_stride, iou_thres, max_det, max_nms, max_wh, agnostic, target, cap, partition,
                       keys, cls_id, fast_scratch, ib, out_boxes, out_count, out_index, nbm);
    EY_LAUNCH_CHECK("ey_nms(resolve)");
    return EY_OK;
  }
  const size_t lds = lds_general;
  if (lds > 60 * 1024 && hipFuncSetAttribute((const void*)nms_select_greedy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return ey_set_error(EY_ELAUNCH, "nms: cannot reserve %zu B of LDS", lds);
  hipLaunchKernelGGL(nms_select_greedy_kernel, dim3(B), dim3(1024), lds, st, nc, A, P, nkeys, multi_label, boxsrc, img_stride, iou_thres, max_det, max_nms, max_wh, agnostic,
                     target, cap, partition, mode, keys, cls_id, out_boxes, out_count, out_index);
  EY_LAUNCH_CHECK("ey_nms(sort_greedy)");
  return EY_OK;
}

static int nms_run(int B, int nc, int A, const float* pred, float conf_thres, float iou_thres, int max_det, int max_nms, float max_wh, int agnostic,
                   int multi_label, const uint8_t* class_mask, float* out_boxes, int32_t* out_count, int32_t* out_index, void* workspace, size_t workspace_bytes,
                   ey_stream_t stream, int mode) {
  EY_CHECK(pred && out_boxes && out_count && workspace, "nms: null pointer");
  EY_CHECK(B > 0 && nc > 0 && A > 0, "nms: bad extent");
  EY_CHECK(max_det > 0 && max_det <= 4096 && max_nms > 0, "nms: max_det=%d (1..4096) max_nms=%d", max_det, max_nms);
  multi_label = multi_label && nc > 1;  // ops.py:240
  EY_CHECK(workspace_bytes >= (multi_label ? ey_nms_workspace_bytes_ml(B, nc, A) : ey_nms_workspace_bytes(B, A)) && ey_aligned(workspace, 8),
           "nms: workspace too small");
  EY_CHECK(!multi_label || (long)A * nc < (1L << 31), "nms: A*nc too large");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* keys = (unsigned long long*)workspace;
  long P;
  int* cls_id = nullptr;
  if (multi_label) {
    P = ((long)A * nc + 255) / 256 * 256;
    const int Apad = (A + 255) / 256 * 256;
    (void)hipMemsetAsync(keys, 0, (size_t)B * P * 8, st);  // (tail padding; every real slot is written by the kernel)
    hipLaunchKernelGGL(nms_score_ml_kernel, dim3((unsigned)((long)Apad * nc / 256), B), dim3(256), 0, st, nc, A, pred, conf_thres, class_mask, keys, P);
  } else {
    P = nms_pow2(A);
    cls_id = (int*)(keys + (size_t)B * P);
    if (A % 4 == 0 && ey_aligned(pred, 16) && ey_aligned(keys, 16) && ey_aligned(cls_id, 16))  // (P is a multiple of 256)
      hipLaunchKernelGGL(nms_score4_kernel, dim3((unsigned)((P / 4 + 255) / 256), B), dim3(256), 0, st, nc, A, pred, conf_thres, class_mask, keys, cls_id, (int)P);
    else
      hipLaunchKernelGGL(nms_score_kernel, dim3((unsigned)(P / 256), B), dim3(256), 0, st, nc, A, pred, conf_thres, class_mask, keys, cls_id, (int)P);
  }
  EY_LAUNCH_CHECK("ey_nms(score)");
  char* fast = multi_label ? nullptr : (char*)workspace + nms_keys_bytes(B, A);
  return nms_select_launch(B, nc, A, P, P, multi_label, pred, (long)(4 + nc) * A, iou_thres, max_det, max_nms, max_wh, agnostic, keys, cls_id, out_boxes, out_count, out_index, st, mode,
                           fast);
}

extern "C" int ey_nms(int B, int nc, int A, const float* pred, float conf_thres, float iou_thres, int max_det, int max_nms, float max_wh, int agnostic,
                      int multi_label, const uint8_t* class_mask, float* out_boxes, int32_t* out_count, int32_t* out_index, void* workspace, size_t workspace_bytes,
                      ey_stream_t stream) {
  EY_CHECK(conf_thres >= 0.f && conf_thres <= 1.f, "nms: Invalid Confidence threshold %f, valid values are between 0.0 and 1.0", conf_thres);
  EY_CHECK(iou_thres >= 0.f && iou_thres <= 1.f, "nms: Invalid IoU %f, valid values are between 0.0 and 1.0", iou_thres);
  return nms_run(B, nc, A, pred, conf_thres, iou_thres, max_det, max_nms, max_wh, agnostic, multi_label, class_mask, out_boxes, out_count, out_index, workspace,
                 workspace_bytes, stream, 0);
}

// ---- Detect.postprocess (head.py:167-189) of the end2end heads: the k = min(max_det, A) best (anchor, class) pairs of every image in
// descending score order, rows [x1,y1,x2,y2,score,class].  The reference takes the k anchors with the best class score and then the k
// best pairs among them; every one of the k best pairs overall belongs to such an anchor (fewer than k pairs, hence fewer than k
// anchors, beat it), so both are the k best pairs overall -- which is the multi-label candidate selection of the NMS kernel (radix
// descent over the A*nc score keys, register / LDS sort) with the suppression switched off.  Equal scores: lower anchor, then lower
// class first (torch.topk leaves that order unspecified).
extern "C" size_t ey_e2e_topk_workspace_bytes(int B, int nc, int A) {
  return (nc > 1 ? ey_nms_workspace_bytes_ml(B, nc, A) : ey_nms_workspace_bytes(B, A)) + (((size_t)B * 4 + 15) & ~(size_t)15);
}
extern "C" int ey_e2e_topk(int B, int nc, int A, const float* pred_xyxy, int k, float* out_rows, int32_t* out_index, void* workspace, size_t workspace_bytes,
                           ey_stream_t stream) {
  EY_CHECK(k > 0 && k <= 4096 && k <= A, "e2e_topk: k=%d (1..min(4096, A=%d))", k, A);
  EY_CHECK(workspace && ey_aligned(workspace, 16) && workspace_bytes >= ey_e2e_topk_workspace_bytes(B, nc, A), "e2e_topk: workspace too small");
  const size_t cnt_bytes = ((size_t)B * 4 + 15) & ~(size_t)15;
  int32_t* count = (int32_t*)workspace;  // (always k: there are A*nc >= k pairs; kept for the shared kernel's interface)
  return nms_run(B, nc, A, pred_xyxy, -1.f, 1.f, k, 0x7fffffff, 0.f, 1, 1, nullptr, out_rows, count, out_index, (char*)workspace + cnt_bytes, workspace_bytes - cnt_bytes,
                 stream, NMS_MODE_XYXY | NMS_MODE_TOPK);
}

extern "C" int ey_nms_candidates(int B, int nc, int A, void* candidates, size_t candidates_bytes, float iou_thres, int max_det, int max_nms, float max_wh,
                                 int agnostic, float* out_boxes, int32_t* out_count, int32_t* out_index, ey_stream_t stream) {
  EY_CHECK(candidates && out_boxes && out_count, "nms: null pointer");
  EY_CHECK(B > 0 && nc > 0 && A > 0, "nms: bad extent");
  EY_CHECK(iou_thres >= 0.f && iou_thres <= 1.f, "nms: Invalid IoU %f, valid values are between 0.0 and 1.0", iou_thres);
  EY_CHECK(max_det > 0 && max_det <= 4096 && max_nms > 0, "nms: max_det=%d (1..4096) max_nms=%d", max_det, max_nms);
  EY_CHECK(candidates_bytes >= ey_nms_candidates_bytes(B, A) && ey_aligned(candidates, 16), "nms: candidate buffer too small");
  const long P = nms_pow2(A);
  const unsigned long long* keys = (const unsigned long long*)candidates;
  const int* cls_id = (const int*)(keys + (size_t)B * P);
  const float* box4 = (const float*)(cls_id + (size_t)B * P);
  // only the A real key slots are scanned (the padding up to P is never written by the fused decode)
  return nms_select_launch(B, nc, A, P, A, 0, box4, 4L * A, iou_thres, max_det, max_nms, max_wh, agnostic, keys, cls_id, out_boxes, out_count, out_index, (hipStream_t)stream, 0,
                           (char*)candidates + nms_cand_bytes(B, A));
}
