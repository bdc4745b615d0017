// Test-time augmentation around the forward path: DetectionModel._predict_augment (ultralytics/nn/tasks.py:372-408).
//   ey_scale_img  = x.flip(3) + scale_img (utils/torch_utils.py:423-432: F.interpolate(bilinear, align_corners=False) to
//                   int(h*r) x int(w*r), then F.pad(value=0.447) to the next stride multiple) in one pass over the planar NCHW image;
//   ey_tta_merge  = _descale_pred (tasks.py:388-397: boxes /= scale, x = img_w - x for the lr flip) + the anchor slice of
//                   _clip_augmented (:399-408) + this scale's share of torch.cat(y, -1).
// Both are HBM-bound copies with a few flops per element (3 launches of each per augmented batch).
#include "common.h"

// ATen's source index for align_corners=False (UpSample.h area_pixel_compute_source_index): max(0, scale*(dst+0.5)-0.5), i1 = i0 + (i0 < in-1),
// weights (1-l, l); value = wy0*(wx0*p00 + wx1*p01) + wy1*(wx0*p10 + wx1*p11) in fp32, one rounding to T at the end.
template <typename T>
__global__ __launch_bounds__(256) void scale_img_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int hs, int ws, int Hp, int Wp, int flip_lr, float sy,
                                                        float sx, float pad) {
  const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
  if (ox >= Wp) return;
  const long plane = blockIdx.z;  // b*C + c
  const T* s = x + plane * H * W;
  float v = pad;
  if (ox < ws && oy < hs) {
    float fy = __fsub_rn(__fmul_rn(sy, (float)oy + 0.5f), 0.5f), fx = __fsub_rn(__fmul_rn(sx, (float)ox + 0.5f), 0.5f);
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = __fsub_rn(fy, (float)y0), lx = __fsub_rn(fx, (float)x0);
    const float wy0 = __fsub_rn(1.f, ly), wx0 = __fsub_rn(1.f, lx);
    // the flip is applied to the SOURCE (the reference resizes x.flip(3)): column c of the flipped image is column W-1-c of x
    const int c0 = flip_lr ? W - 1 - x0 : x0, c1 = flip_lr ? W - 1 - x1 : x1;
    const float p00 = to_f(s[(long)y0 * W + c0]), p01 = to_f(s[(long)y0 * W + c1]), p10 = to_f(s[(long)y1 * W + c0]), p11 = to_f(s[(long)y1 * W + c1]);
    const float r0 = __fadd_rn(__fmul_rn(wx0, p00), __fmul_rn(lx, p01)), r1 = __fadd_rn(__fmul_rn(wx0, p10), __fmul_rn(lx, p11));
    v = __fadd_rn(__fmul_rn(wy0, r0), __fmul_rn(ly, r1));
  }
  y[plane * Hp * Wp + (long)oy * Wp + ox] = from_f<T>(v);
}

extern "C" int ey_scale_img(int dtype, int B, int C, int H, int W, const void* x, int hs, int ws, int Hp, int Wp, int flip_lr, float pad_value, void* y,
                            ey_stream_t stream) {
  EY_CHECK(x && y && x != y, "scale_img: null / aliased pointers");
  EY_CHECK(B > 0 && C > 0 && (long)B * C <= 65535 && H > 0 && W > 0, "scale_img: bad extent (%d,%d,%d,%d)", B, C, H, W);
  EY_CHECK(hs > 0 && ws > 0 && hs <= Hp && ws <= Wp && Hp <= 65535, "scale_img: resized %dx%d does not fit the padded %dx%d", hs, ws, Hp, Wp);
  const float sy = (float)H / (float)hs, sx = (float)W / (float)ws;  // area_pixel_compute_scale<float>(in, out, align_corners=false, nullopt)
  dim3 grid((Wp + 255) / 256, Hp, B * C);
  if (dtype == EY_F16)
    hipLaunchKernelGGL(scale_img_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, H, W, hs, ws, Hp, Wp, flip_lr, sy, sx, pad_value);
  else if (dtype == EY_F32)
    hipLaunchKernelGGL(scale_img_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, H, W, hs, ws, Hp, Wp, flip_lr, sy, sx, pad_value);
  else
    return ey_set_error(EY_EINVAL, "scale_img: bad dtype");
  EY_LAUNCH_CHECK("ey_scale_img");
  return EY_OK;
}

// pred (B, no, A) fp32 -> out (B, no, A_out) fp32, columns [out_off, out_off + hi - lo) <- anchors [lo, hi) of pred
__global__ __launch_bounds__(256) void tta_merge_kernel(const float* __restrict__ p, float* __restrict__ out, int no, int A, int lo, int n, long A_out, int out_off,
                                                        float scale, int flip, float img_h, float img_w) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= n) return;
  const int r = blockIdx.y, b = blockIdx.z;
  float v = p[((long)b * no + r) * A + lo + a];
  if (r < 4) {
    v = __fdiv_rn(v, scale);  // p[:, :4] /= scale
    if (r == 0 && flip == 3) v = __fsub_rn(img_w, v);
    if (r == 1 && flip == 2) v = __fsub_rn(img_h, v);
  }
  out[((long)b * no + r) * A_out + out_off + a] = v;
}

extern "C" int ey_tta_merge(int B, int no, int A, const float* pred, int lo, int hi, float scale, int flip, int img_h, int img_w, float* out, long A_out, int out_off,
                            ey_stream_t stream) {
  EY_CHECK(pred && out, "tta_merge: null pointer");
  EY_CHECK(B > 0 && B <= 65535 && no >= 4 && no <= 65535 && A > 0, "tta_merge: bad extent");
  EY_CHECK(lo >= 0 && lo <= hi && hi <= A && out_off >= 0 && (long)out_off + (hi - lo) <= A_out, "tta_merge: anchors [%d,%d) of %d -> columns %d.. of %ld", lo, hi, A,
           out_off, A_out);
  EY_CHECK(scale > 0.f && (flip == 0 || flip == 2 || flip == 3), "tta_merge: scale %g flip %d", scale, flip);
  if (hi == lo) return EY_OK;
  dim3 grid((hi - lo + 255) / 256, no, B);
  hipLaunchKernelGGL(tta_merge_kernel, grid, dim3(256), 0, (hipStream_t)stream, pred, out, no, A, lo, hi - lo, A_out, out_off, scale, flip, (float)img_h, (float)img_w);
  EY_LAUNCH_CHECK("ey_tta_merge");
  return EY_OK;
}
