// K9+K10 head decode (DGQP quality + DFL expectation + anchor decode + score modulation) and K11 batched NMS.
#include "common.h"

// ============================================================================ head decode, one pyramid level
// thread = one anchor.  64 box logits -> 4 x softmax(16) -> {DFL expectation, top-4 + mean -> FC(20->hid, ReLU)
// -> FC(hid->1, sigmoid)} ; scores = sigmoid(cls) * clamp(q, 1e-6, 1-1e-6) ; boxes = xywh * stride.  All fp32.
template <typename T>
__global__ __launch_bounds__(256) void head_decode_kernel(int B, int H, int W, int nc, float stride, const T* __restrict__ box, int boxCs,
                                                          const T* __restrict__ cls, int clsCs, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, int hid, float* __restrict__ pred, int A, int a_off) {
  const int HW = H * W;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * HW) return;
  const int b = (int)(idx / HW), a = (int)(idx - (long)b * HW);
  const int ay = a / W, ax = a - ay * W;
  const T* bp = box + idx * boxCs;
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
    for (int j = 0; j < hid; ++j) {
      float hsum = b1[j];
#pragma unroll
      for (int i = 0; i < 20; ++i) hsum += w1[j * 20 + i] * stat[i];
      o += w2[j] * fmaxf(hsum, 0.f);
    }
    q = fminf(fmaxf(ey_sigmoid(o), 1e-6f), 1.f - 1e-6f);
  }
  const float cx0 = ax + 0.5f, cy0 = ay + 0.5f;
  const float x1 = cx0 - dist[0], y1 = cy0 - dist[1], x2 = cx0 + dist[2], y2 = cy0 + dist[3];
  float* pp = pred + (long)b * (4 + nc) * A + a_off + a;
  pp[0] = (x1 + x2) * 0.5f * stride;
  pp[(long)A] = (y1 + y2) * 0.5f * stride;
  pp[2L * A] = (x2 - x1) * stride;
  pp[3L * A] = (y2 - y1) * stride;
  const T* cp = cls + idx * clsCs;
  for (int c = 0; c < nc; ++c) pp[(long)(4 + c) * A] = ey_sigmoid(to_f(cp[c])) * q;
}

extern "C" int ey_head_decode(int dtype, int B, int H, int W, int nc, float stride, const void* box, int box_cstride, const void* cls,
                              int cls_cstride, const float* q_w1, const float* q_b1, const float* q_w2, const float* q_b2, int q_hidden,
                              float* pred, int A_total, int a_off, ey_stream_t stream) {
  EY_CHECK(box && cls && pred, "head_decode: null pointer");
  EY_CHECK(dtype == EY_F16 || dtype == EY_F32, "head_decode: bad dtype");
  EY_CHECK(B > 0 && H > 0 && W > 0 && nc > 0, "head_decode: bad extent");
  EY_CHECK(box_cstride >= 64 && cls_cstride >= nc, "head_decode: cstride");
  EY_CHECK((box_cstride * (dtype == EY_F16 ? 2 : 4)) % 16 == 0 && ey_aligned(box, 16), "head_decode: box view must be 16-byte aligned");
  EY_CHECK(a_off >= 0 && a_off + H * W <= A_total, "head_decode: level [%d,%d) outside A=%d", a_off, a_off + H * W, A_total);
  EY_CHECK(!q_w1 || (q_b1 && q_w2 && q_b2 && q_hidden > 0), "head_decode: incomplete quality head");
  const long total = (long)B * H * W;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == EY_F16)
    hipLaunchKernelGGL(head_decode_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, B, H, W, nc, stride, (const f16*)box, box_cstride, (const f16*)cls,
                       cls_cstride, q_w1, q_b1, q_w2, q_b2, q_hidden, pred, A_total, a_off);
  else
    hipLaunchKernelGGL(head_decode_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, B, H, W, nc, stride, (const float*)box, box_cstride,
                       (const float*)cls, cls_cstride, q_w1, q_b1, q_w2, q_b2, q_hidden, pred, A_total, a_off);
  EY_LAUNCH_CHECK("ey_head_decode");
  return EY_OK;
}

// ============================================================================ NMS
// Stage 1 (nms_score): per anchor best class (first maximal index, ops.py:274) and confidence; anchors that pass
//   conf > thr (and the class filter) get the 64-bit key  (score_bits << 32) | (0xFFFFFFFF - anchor); others 0.
//   Sorting keys DESCENDING = scores descending with ties broken by ascending candidate (= anchor) order, i.e. the
//   stable descending sort torchvision.ops.nms applies to the compacted candidate list (ops.py:253,275,296).
// Stage 2 (nms_sort_greedy): one workgroup per image: bitonic sort of the keys (LDS when <= 16384 keys, else in
//   the global workspace), then greedy suppression by ONE wave: 64 sorted candidates at a time are tested against
//   the kept list (LDS) and then resolved inside the wave with ballots.  Stops at max_det kept or at the first
//   zero key.  IoU arithmetic is the torchvision CPU kernel's, op by op, with explicit round-to-nearest intrinsics
//   (no FMA contraction) so that decisions are bit-identical with the fp32 CPU oracle.
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

struct KeptBox { float x1, y1, x2, y2, area; };

__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1, float bx2, float by2,
                                       float barea, float thr) {
  const float xx1 = fmaxf(ax1, bx1), yy1 = fmaxf(ay1, by1), xx2 = fminf(ax2, bx2), yy2 = fminf(ay2, by2);
  const float w = fmaxf(0.f, __fsub_rn(xx2, xx1)), h = fmaxf(0.f, __fsub_rn(yy2, yy1));
  const float inter = __fmul_rn(w, h);
  const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(aarea, barea), inter));
  return ovr > thr;
}

template <bool LDS_SORT>
__global__ __launch_bounds__(1024) void nms_sort_greedy_kernel(int nc, int A, int P, const float* __restrict__ pred, float iou_thres, int max_det,
                                                               int max_nms, float max_wh, int agnostic, unsigned long long* __restrict__ gkeys,
                                                               const int* __restrict__ cls_id, float* __restrict__ out_boxes, int* __restrict__ out_count, int* __restrict__ out_index) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  unsigned long long* keys = LDS_SORT ? reinterpret_cast<unsigned long long*>(smem) : gkeys + (long)b * P;
  KeptBox* kept = reinterpret_cast<KeptBox*>(smem + (LDS_SORT ? (size_t)P * 8 : 0));
  if (LDS_SORT) {
    for (int i = threadIdx.x; i < P; i += blockDim.x) keys[i] = gkeys[(long)b * P + i];
  }
  __syncthreads();
  // bitonic sort, descending
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long u = keys[i], v = keys[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (u < v) : (u > v)) { keys[i] = v; keys[ixj] = u; }
        }
      }
      __syncthreads();
    }
  }
  if (threadIdx.x >= 64) return;  // greedy part: one wave (no further block barriers below)
  const int lane = threadIdx.x;
  const float* pb = pred + (long)b * (4 + nc) * A;
  int nkept = 0;
  const int limit = min(P, max_nms);
  for (int base = 0; base < limit && nkept < max_det; base += 64) {
    const int i = base + lane;
    const unsigned long long key = i < limit ? keys[i] : 0ull;
    bool alive = key != 0ull;
    if (__ballot(alive) == 0ull) break;  // sorted: nothing but zeros from here on
    float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, area = 0.f, conf = 0.f, ux1 = 0.f, uy1 = 0.f, ux2 = 0.f, uy2 = 0.f;
    int a = 0, ci = 0;
    if (alive) {
      a = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
      conf = __uint_as_float((unsigned)(key >> 32));
      const float cx = pb[a], cy = pb[(long)A + a], w = pb[2L * A + a], h = pb[3L * A + a];
      const float hw = __fmul_rn(w, 0.5f), hh = __fmul_rn(h, 0.5f);  // xywh2xyxy, ops.py:430-432 (x/2 is exact)
      ux1 = __fsub_rn(cx, hw); uy1 = __fsub_rn(cy, hh); ux2 = __fadd_rn(cx, hw); uy2 = __fadd_rn(cy, hh);
      ci = cls_id[(long)b * P + a];
      const float off = agnostic ? 0.f : __fmul_rn((float)ci, max_wh);  // ops.py:289
      x1 = __fadd_rn(ux1, off); y1 = __fadd_rn(uy1, off); x2 = __fadd_rn(ux2, off); y2 = __fadd_rn(uy2, off);
      area = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
      for (int k = 0; k < nkept && alive; ++k) {
        const KeptBox kb = kept[k];
        if (iou_gt(kb.x1, kb.y1, kb.x2, kb.y2, kb.area, x1, y1, x2, y2, area, iou_thres)) alive = false;
      }
    }
    // resolve inside the wave, in score order
    for (int s = 0; s < 64; ++s) {
      const unsigned long long am = __ballot(alive);
      if (!((am >> s) & 1ull)) continue;
      if (nkept >= max_det) break;
      const float sx1 = __shfl(x1, s), sy1 = __shfl(y1, s), sx2 = __shfl(x2, s), sy2 = __shfl(y2, s), sarea = __shfl(area, s);
      if (lane == s) {
        kept[nkept] = KeptBox{x1, y1, x2, y2, area};
        float* ob = out_boxes + ((long)b * max_det + nkept) * 6;
        ob[0] = ux1; ob[1] = uy1; ob[2] = ux2; ob[3] = uy2; ob[4] = conf; ob[5] = (float)ci;
        if (out_index) out_index[(long)b * max_det + nkept] = a;
      } else if (lane > s && alive) {
        if (iou_gt(sx1, sy1, sx2, sy2, sarea, x1, y1, x2, y2, area, iou_thres)) alive = false;
      }
      ++nkept;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) out_count[b] = nkept;
  // zero the unused tail rows so the output is deterministic
  for (int r = nkept * 6 + lane; r < max_det * 6; r += 64) out_boxes[(long)b * max_det * 6 + r] = 0.f;
  if (out_index)
    for (int r = nkept + lane; r < max_det; r += 64) out_index[(long)b * max_det + r] = -1;
}

static int nms_pow2(int A) { int p = 64; while (p < A) p <<= 1; return p; }

extern "C" size_t ey_nms_workspace_bytes(int B, int A) { return (size_t)B * nms_pow2(A) * (8 + 4); }  // keys + class ids

extern "C" int ey_nms(int B, int nc, int A, const float* pred, float conf_thres, float iou_thres, int max_det, int max_nms, float max_wh, int agnostic,
                      const uint8_t* class_mask, float* out_boxes, int32_t* out_count, int32_t* out_index, void* workspace, size_t workspace_bytes,
                      ey_stream_t stream) {
  EY_CHECK(pred && out_boxes && out_count && workspace, "nms: null pointer");
  EY_CHECK(B > 0 && nc > 0 && A > 0, "nms: bad extent");
  EY_CHECK(conf_thres >= 0.f && conf_thres <= 1.f, "nms: Invalid Confidence threshold %f, valid values are between 0.0 and 1.0", conf_thres);
  EY_CHECK(iou_thres >= 0.f && iou_thres <= 1.f, "nms: Invalid IoU %f, valid values are between 0.0 and 1.0", iou_thres);
  EY_CHECK(max_det > 0 && max_det <= 4096 && max_nms > 0, "nms: max_det=%d (1..4096) max_nms=%d", max_det, max_nms);
  EY_CHECK(workspace_bytes >= ey_nms_workspace_bytes(B, A) && ey_aligned(workspace, 8), "nms: workspace too small");
  const int P = nms_pow2(A);
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* keys = (unsigned long long*)workspace;
  int* cls_id = (int*)(keys + (size_t)B * P);
  hipLaunchKernelGGL(nms_score_kernel, dim3(P / 64 >= 4 ? P / 256 : 1, B), dim3(P / 64 >= 4 ? 256 : P), 0, st, nc, A, pred, conf_thres, class_mask,
                     keys, cls_id, P);
  EY_LAUNCH_CHECK("ey_nms(score)");
  const size_t kept_bytes = (size_t)max_det * sizeof(KeptBox);
  if ((size_t)P * 8 + kept_bytes <= 160 * 1024) {
    const size_t lds = (size_t)P * 8 + kept_bytes;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)nms_sort_greedy_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ey_set_error(EY_ELAUNCH, "nms: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(nms_sort_greedy_kernel<true>, dim3(B), dim3(1024), lds, st, nc, A, P, pred, iou_thres, max_det, max_nms, max_wh, agnostic,
                       keys, cls_id, out_boxes, out_count, out_index);
  } else {
    hipLaunchKernelGGL(nms_sort_greedy_kernel<false>, dim3(B), dim3(1024), kept_bytes, st, nc, A, P, pred, iou_thres, max_det, max_nms, max_wh, agnostic,
                       keys, cls_id, out_boxes, out_count, out_index);
  }
  EY_LAUNCH_CHECK("ey_nms(sort_greedy)");
  return EY_OK;
}
