/*
 * edgeyolo_hip.h — C ABI of libedgeyolo_hip.so: the MI355X (gfx950) detection forward path of EdgeLine-YOLO.
 *
 * The reference (OneWalkman/EDGE-YOLO, an Ultralytics 8.3.63 fork) is pure Python/PyTorch and has NO FFI of its
 * own; the operators below are the ATen / torchvision call sites of its predict() hot path.
 * Each entry point names the reference function it replaces (paths relative to /root/reference/ultralytics/).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory unless it says "host";
 *  - the caller owns all buffers; nothing is allocated, freed or synchronised inside; every launch goes to the
 *    hipStream_t passed as `stream` (so the calls can be captured into a hipGraph);
 *  - return 0 on success, a negative EY_E* code otherwise; ey_last_error() gives the message (thread local);
 *  - activations are NHWC ("channels last"): element (b,y,x,c) of a view lives at
 *        ptr + (((b*H + y)*W + x) * cstride + c)        [elements],
 *    so a channel slice of a wider tensor is just (ptr + c0, cstride = C_total): chunk/split/cat never copy;
 *  - dtype is the storage type of activations and packed weights (EY_F16 | EY_F32); accumulation, bias,
 *    softmax, decode and NMS arithmetic are always fp32.  EY_F32 is the parity mode (exact-f32 MFMA),
 *    EY_F16 the throughput mode.
 */
#ifndef EDGEYOLO_HIP_H
#define EDGEYOLO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ey_stream_t; /* hipStream_t */

enum { EY_F16 = 0, EY_F32 = 1 };
enum { EY_ACT_NONE = 0, EY_ACT_SILU = 1, EY_ACT_RELU = 2, EY_ACT_SIGMOID = 3 };
enum { EY_OK = 0, EY_EINVAL = -1, EY_EUNSUPPORTED = -2, EY_ELAUNCH = -3 };

const char* ey_last_error(void);
int ey_version(void);
/* sizeof(ey_conv_desc) (which=0) / sizeof(ey_conv_direct_desc) (which=1) as compiled: lets a binding check its struct layout. */
size_t ey_abi_sizeof(int which);
/* Dispatch tunables (developer tools only; csrc/tune.h lists the names and the measured defaults).  They choose between kernels
 * that compute the same result -- thresholds such as "lean pointwise kernel below pw_m output pixels" -- never the arithmetic.
 * The library reads NO environment variables.  ey_tune_set returns EY_EINVAL for an unknown name; ey_tune_get returns -1.
 * Process-wide, not synchronised: set them before the first launch. */
int ey_tune_set(const char* name, long value);
long ey_tune_get(const char* name);

/* ---- K1: dense convolution (+ folded-BN bias, activation, residual) as NHWC implicit GEMM on MFMA ------------
 * Replaces Conv.forward_fuse (nn/modules/conv.py:57-59, BN folded by utils/torch_utils.py:238-265), the raw
 * nn.Conv2d+bias calls of the heads (nn/modules/head.py:61,70) and LinearAttention.qkv/proj (block.py:3357-3358);
 * also absorbs nn.Upsample(nearest x2)+Concat feeding a conv (yaml layers 11-12,14-15,18,21; conv.py:353-355),
 * Bottleneck/PSA residual adds (block.py:480,3446-3449) and the _WaveletEnhancer tail (block.py:3706-3710).
 *
 *   y = res + out_scale * act( conv_k(cat_c(src0, src1)) + bias + bilinear_up2x(addz) )
 *
 * src[i] may be read through a nearest x2^up upsample (src_up = 0|1).  k in {1,3}, stride in {1,2}, pad = k/2.
 * `ngroup` > 1 runs over `ngroup` channel-offset slices (src0 += g*src_gstride, y += g*y_gstride elements) with
 * weight set min(g, w_gmax) — the three high-frequency sub-bands share f_h, and f_ll (a 1x1 conv written as a
 * centre-tap 3x3) rides in the same launch as group 0 (block.py:3688-3694).
 * Weights must be packed by ey_conv_pack_weight(). */
typedef struct {
  int32_t dtype;
  int32_t B, H, W;   /* logical conv input extent (after the virtual upsample) */
  int32_t Ho, Wo;    /* output extent */
  int32_t Cout;
  int32_t k, stride, pad;
  int32_t act;
  int32_t nsrc;      /* 1 or 2 */
  const void* src[2];
  int32_t src_C[2];
  int32_t src_cstride[2];
  int32_t src_up[2];
  const void* w;     /* packed, see ey_conv_pack_weight */
  const float* bias; /* [Cout] fp32 or NULL */
  void* y;
  int32_t y_cstride;
  const void* res;   /* optional, [B,Ho,Wo,Cout] view, same dtype */
  int32_t res_cstride;
  float out_scale;   /* 1.0f unless the wavelet tail (tanh(gamma)) */
  const void* addz;  /* optional, [B,addz_H,addz_W,Cout] view, bilinearly resized (align_corners=False) to Ho x Wo */
  int32_t addz_cstride;
  int32_t addz_H, addz_W;
  int32_t ngroup;
  int64_t src_gstride, y_gstride;
  /* group g uses weight set min(g, w_gmax): w + that*w_gstride (elements of dtype), bias + that*Cout.  0/0 = shared. */
  int64_t w_gstride;
  int32_t w_gmax;
} ey_conv_desc;

/* Bytes of the packed weight buffer for a conv with Cout x (k*k*Cin) (Cin = sum of src_C). */
size_t ey_conv_packed_bytes(int dtype, int Cout, int Cin, int k);
/* host -> host: w_oihw fp32 [Cout][Cin][k][k]  ->  packed rows [Cout_pad][k][k][Cin] (+ row permutation for the
 * MFMA epilogue, + zero slack).  Upload `out` to the device afterwards. */
int ey_conv_pack_weight(int dtype, int Cout, int Cin, int k, const float* w_oihw_host, void* out_host, size_t out_bytes);
int ey_conv2d(const ey_conv_desc* d, ey_stream_t stream);
/* Kernel instantiation ey_conv2d launches for a shape (profiling only): kind*1000 + NT*10 + MT, kind 3 =
 * conv_small_kernel<T,NT,BATCH> (small-M latency kernel), 2 = conv3_halo_kernel<T,NT,stride> (3x3, LDS halo tile),
 * 1 = conv_ws_kernel<T,NT,MT,k> (weight-stationary persistent), 0 = conv_igemm_kernel<T,NT,MT> (K-chunked fallback).
 * plain_single_source = one source, no upsample.  Shapes taken by the pointwise / tile / register-stationary kernels are
 * reported by ey_conv_last_variant() after the launch. */
int ey_conv_variant(int dtype, int Cout, int Cin, int k, int stride, int plain_single_source, long M, int ngroup);
/* Code of the kernel the last ey_conv2d on this thread launched when ey_conv_variant() does not describe it (else 0):
 * 4000 + NT*10 + nsrc = conv_pw_kernel<T,NT,...> (lean pointwise kernel for small maps); 5000 + NT*10 + KS =
 * conv_pwr_kernel<T,NT,KS,...> (register-stationary pointwise kernel for large maps); 6000 + NT*10 + stride =
 * conv3_tile_kernel<T,NT,S,...> (3x3 LDS tile kernel); 7000 + NT*10 + stride = conv3r_kernel<NT,S> (register-stationary
 * 3x3 kernel for Cin = 16).  Profiling labels only. */
int ey_conv_last_variant(void);
/* ---- two chained pointwise convs in one kernel, registers only: y = act2(W2 . act1(W1 . x + b1) + b2) -- the last two 1x1 convs
 * of the Detect class tower (Conv(c3,c3,1)+SiLU, nn.Conv2d(c3,nc,1); head.py:68-70).  w1_packed = ey_conv_pack_weight(Cmid, Cin, 1);
 * w2_packed = ey_conv_pack_weight(Cout, ey_conv_chain_klen(Cmid), 1) of W2 with its INPUT columns gathered by ey_conv_chain_kperm
 * (perm[k'] = mid channel feeding k-slot k', -1 = zero column): the second contraction runs in the order the first GEMM leaves its
 * results in the registers.  f16 only; built for the two class-tower shapes Cin 72..96 -> Cmid 80 -> Cout <= 80 (nc = 80) and Cin 40..64 ->
 * Cmid 64 -> Cout <= 16 (small class counts, e.g. GC10-DET nc = 10); EY_EUNSUPPORTED otherwise (run the two convs). */
int ey_conv_chain_klen(int Cmid);
int ey_conv_chain_kperm(int Cmid, int* perm_host, int perm_len);
int ey_conv_pw_chain(int dtype, int B, int H, int W, int Cin, int Cmid, int Cout, const void* x, int x_cstride, const void* w1_packed,
                     const float* b1, int act1, const void* w2_packed, const float* b2, int act2, void* y, int y_cstride, ey_stream_t stream);
/* NT the weights of a Cout-channel conv are packed with (row permutation of ey_conv_pack_weight). */
int ey_conv_pack_nt(int Cout);

/* ---- generic direct convolution (any Cin/Cout/groups/k/stride; scalar) — correctness path for shapes the MFMA
 * kernel does not take (channel counts not multiples of 8, grouped convs).  Weights: fp32 OIHW on device. */
typedef struct {
  int32_t dtype;
  int32_t B, H, W, Cin, Ho, Wo, Cout;
  int32_t k, stride, pad, groups, act;
  const void* x; int32_t x_cstride;
  const float* w_oihw; const float* bias;
  void* y; int32_t y_cstride;
} ey_conv_direct_desc;
int ey_conv2d_direct(const ey_conv_direct_desc* d, ey_stream_t stream);

/* ---- stem: 3x3 stride-2 conv on the NCHW input image -> NHWC, + bias + SiLU (layer 0; conv.py:41-59).
 * x: [B,Cin,H,W] contiguous, x_dtype EY_F16|EY_F32; w: fp32 [Cout][Cin][3][3] device; Cin<=4, Cout%16==0. */
int ey_stem_conv(int x_dtype, int y_dtype, int B, int Cin, int H, int W, int Cout, int act, const void* x_nchw,
                 const float* w_oihw, const float* bias, void* y, int y_cstride, ey_stream_t stream);

/* ---- two chained 1x1 convs as one launch (f16, 64 or 128 channels, small maps): `second` must read exactly what `first` writes
 * (second->src[0] == first->y, same cstride, one source each; `first` may carry bias / addz / activation / out_scale / residual, `second`
 * bias / activation).  Built for the _WaveletEnhancer tail conv (block.py:3700-3710) + the stacked cv1|cv2 conv of the DSC3k behind it
 * (block.py:382-396).  Both outputs are written; bit-identical to two ey_conv2d calls.  EY_EUNSUPPORTED (nothing launched) otherwise. */
int ey_conv_pw_pair(const ey_conv_desc* first, const ey_conv_desc* second, ey_stream_t stream);

/* ---- a block's closing 1x1 conv over a two-part virtual concat + the stride-2 3x3 conv behind it, as one kernel (f16; Cmid = Cout = 64;
 * C0, C1 <= 32, multiples of 8; both bias + SiLU):   y = act2( Conv3x3s2( act1( Conv1x1( cat(src0, src1) ) + bias1 ) ) + bias2 )
 * = DSC3K2_Wavelet.cv2 / C2f.cv2 (block.py:357-396,3783-3788) + the next backbone layer Conv(64,64,3,2) (conv.py:41-59).  The (B,64,H,W)
 * map between them stays in LDS, rounded to f16 as ey_conv2d stores it: bit-identical to the two ey_conv2d calls.  w1: ey_conv_pack_weight
 * (EY_F16, 64, C0 + C1, 1), w2: (EY_F16, 64, 64, 3).  EY_EUNSUPPORTED (before anything is launched) for every other shape. */
int ey_conv_pw_conv3s2(int dtype, int B, int H, int W, const void* src0, int C0, int cstride0, const void* src1, int C1, int cstride1, int Cmid,
                       const void* w1_packed, const float* bias1, int act1, int Cout, const void* w2_packed, const float* bias2, int act2, void* y,
                       int y_cstride, ey_stream_t stream);

/* ---- layers 0 + 1 as one kernel (f16): y = act1( Conv3x3s2_{16->C1}( act0( Conv3x3s2_{3->16}(x) + bias0 ) ) + bias1 ), conv.py:41-59 twice.
 * x: [B,3,H,W] contiguous f16 (W % 8 == 0); w0: fp32 [16][3][3][3] device; w1: ey_conv_pack_weight(EY_F16, C1, 16, 3, ...); C1 = 32.
 * The (B,16,H/2,W/2) intermediate stays in LDS (rounded to f16 as ey_stem_conv stores it): bit-identical to ey_stem_conv + ey_conv2d.
 * EY_EUNSUPPORTED (before anything is launched) for every other shape. */
int ey_stem_pair(int B, int H, int W, const void* x_nchw, const float* w0_oihw, const float* bias0, int act0, int C1, const void* w1_packed,
                 const float* bias1, int act1, void* y, int y_cstride, ey_stream_t stream);

/* ---- K2/K3: depthwise kxk, stride 1, pad k/2 (+ optional bias + activation).  DSConv.dw (conv.py:94-97,102) and
 * DWConv (conv.py:124-129) in Detect.cv3 (head.py:68-69).  w: [k][k][C] in `dtype`; k in {3,5,7}; C%8==0. */
int ey_dwconv(int dtype, int B, int H, int W, int C, int k, int act, const void* x, int x_cstride, const void* w_kkc,
              const float* bias, void* y, int y_cstride, ey_stream_t stream);

/* ---- K2/K3 fused: depthwise kxk -> pointwise 1x1 in one kernel:
 *   y = res + act( pw1x1( dw_act( dw_kxk(x) + dw_bias ) ) + bias )
 * = DSConv.forward (conv.py:101-104; dw_bias NULL, dw_act NONE, BN folded into pw) and the Detect.cv3 pairs
 * DWConv(BN+SiLU) -> Conv 1x1 (head.py:68-69).  The depthwise result stays in LDS (rounded to `dtype` like the
 * reference's intermediate tensor).  w_dw: [k][k][Cin] in `dtype`; dw_bias fp32 [Cin] or NULL;
 * w_pw: ey_conv_pack_weight(dtype, Cout, Cin, 1, ...); k in {3,5,7}; Cin%8==0, Cin<=256. */
int ey_dsconv(int dtype, int B, int H, int W, int Cin, int Cout, int k, int act, const void* x, int x_cstride,
              const void* w_dw_kkc, const float* dw_bias, int dw_act, const void* w_pw_packed, const float* bias, void* y, int y_cstride, const void* res,
              int res_cstride, ey_stream_t stream);

/* Toeplitz form of the depthwise stage for wide kernels (k = 7; f16; Cin 16 or 32, Cout <= 32): the row filter becomes one
 * 16x16x32 MFMA per channel and filter row (7 MFMAs per channel per 16x16 tile instead of 12544 FMAs).  w_dw_toeplitz is
 * built once on the host by ey_dsconv_pack_toeplitz from the fp32 [k][k][C] depthwise weights (ey_dsconv_toeplitz_bytes bytes,
 * then uploaded).  Same arguments and result as ey_dsconv; shapes outside the Toeplitz kernel are forwarded to ey_dsconv. */
size_t ey_dsconv_toeplitz_bytes(int C, int k);
int ey_dsconv_pack_toeplitz(int C, int k, const float* w_dw_kkc_host, void* out_host, size_t out_bytes);
int ey_dsconv_tz(int dtype, int B, int H, int W, int Cin, int Cout, int k, int act, const void* x, int x_cstride,
                 const void* w_dw_kkc, const void* w_dw_toeplitz, const float* dw_bias, int dw_act, const void* w_pw_packed, const float* bias,
                 void* y, int y_cstride, const void* res, int res_cstride, ey_stream_t stream);

/* ---- K2 pair: DSBottleneck.forward (block.py:1496-1503) as one kernel on small maps (f16; C = c1 = c_ = c2 in {32, 64}; k1 = 3,
 * k2 in {5, 7}; stride 1):   y = [x +] DSConv_k2( DSConv_k1(x) ),  DSConv_k(t) = act( pw1x1( dw_kxk(t) ) + bias ).
 * A workgroup owns a band of rows of one image and recomputes the first DSConv on the second one's halo rows; the intermediate
 * tensor stays in LDS, rounded to f16 exactly as the two-launch form (2 x ey_dsconv) stores it: the result is bit-identical to that
 * form.  Weights as for ey_dsconv (no depthwise bias).  Returns EY_EUNSUPPORTED -- before anything is launched -- for every other
 * shape; the caller then issues the two ey_dsconv calls. */
int ey_dsb_pair(int dtype, int B, int H, int W, int C, int k1, int k2, int act, const void* x, int x_cstride, const void* w_dw1_kkc,
                const void* w_pw1_packed, const float* bias1, const void* w_dw2_kkc, const void* w_pw2_packed, const float* bias2, int add_residual,
                void* y, int y_cstride, ey_stream_t stream);

/* Kernel the last ey_dsconv / ey_dsconv_tz on this thread launched (profiling labels): 1 = dsconv_kernel (LDS tile),
 * 2 = dsconv_strip_kernel (register strip), 3 = dsconv_tz_kernel (Toeplitz MFMA). */
int ey_dsconv_last_variant(void);

/* ---- K4: single-level 2-D Haar analysis (_PywtDWT2D.forward, block.py:3619-3642).
 * x [B,H,W,C] -> y [B,H/2,W/2,4C] with channel blocks LL|LH|HL|HH; odd H/W floor like the stride-2 conv. */
int ey_dwt_haar(int dtype, int B, int H, int W, int C, const void* x, int x_cstride, void* y, int y_cstride,
                ey_stream_t stream);

/* ---- K4+K5 fused: the half-resolution branch of _WaveletEnhancer (block.py:3688-3706) in one kernel (f16 only):
 *   Z = W_z . cat[ SiLU(f_ll(LL)+b), SiLU(f_h(LH)+b), SiLU(f_h(HL)+b), SiLU(f_h(HH)+b) ]   with (LL,LH,HL,HH) = Haar DWT of x
 * x [B,H,W,C] -> z [B,H/2,W/2,C].  w_sub_packed: two ey_conv_pack_weight(EY_F16, C/2, C, 3) sets w_set_stride ELEMENTS apart (set 0 =
 * f_ll written as a centre-tap 3x3, set 1 = the shared f_h); b_sub fp32 [2][C/2]; w_z_packed = ey_conv_pack_weight(EY_F16, C, 2C, 1)
 * (the fuse conv's columns over the processed sub-bands with the band weights folded in).  C in {16,32,64,128}; EY_EUNSUPPORTED
 * otherwise / for fp32 (run ey_dwt_haar + ey_conv2d x2).  Replaces three launches and the HBM round trip of the 4C-channel sub-band
 * tensor and the 2C-channel P. */
int ey_wavelet_z(int dtype, int B, int H, int W, int C, const void* x, int x_cstride, const void* w_sub_packed, long w_set_stride, const float* b_sub,
                 const void* w_z_packed, void* z, int z_cstride, ey_stream_t stream);

/* ---- K6: the three chained 5x5/s1/p2 max-pools of SPPF (block.py:219-223): y1=mp(x), y2=mp(y1), y3=mp(y2). */
int ey_sppf_pool(int dtype, int B, int H, int W, int C, const void* x, int x_cstride, void* y1, void* y2, void* y3,
                 int y_cstride, ey_stream_t stream);

/* ---- K7 (module-level form): channel-slice copy with optional nearest x2 upsample — nn.Upsample / Concat
 * (conv.py:345-355) when they are not folded into the consuming conv.  dst[b,y,x,c] = src[b,y>>up,x>>up,c]. */
int ey_copy_nhwc(int dtype, int B, int H, int W, int C, int up, const void* src, int src_cstride, void* dst,
                 int dst_cstride, ey_stream_t stream);
/* NCHW-contiguous <-> NHWC view transposes (boundary with callers that hand over plain contiguous tensors). */
int ey_nchw_to_nhwc(int dtype, int B, int C, int H, int W, const void* src, void* dst, int dst_cstride, ey_stream_t stream);
int ey_nhwc_to_nchw(int dtype, int B, int C, int H, int W, const void* src, int src_cstride, void* dst, ey_stream_t stream);

/* ---- GPU pre-processing: LetterBox.__call__ (data/augment.py:1556-1591: cv2.resize INTER_LINEAR to
 * (new_h,new_w) + copyMakeBorder with 114) and BasePredictor.preprocess (engine/predictor.py:123-133: BGR->RGB,
 * HWC->CHW, uint8 -> f16/f32, /255) for ONE image, as one kernel.  src: device uint8 [src_h][src_w][3] (row pitch
 * src_row_bytes); dst: one [3][H][W] image slot of the NCHW batch tensor (out_dtype).  The caller computes the
 * LetterBox geometry (new size, top/left padding); the resize follows OpenCV's 8-bit fixed-point INTER_LINEAR path. */
int ey_letterbox(int out_dtype, const uint8_t* src_hwc, int src_h, int src_w, int src_row_bytes, void* dst_chw, int H, int W,
                 int new_h, int new_w, int top, int left, int pad_value, int swap_rb, ey_stream_t stream);
/* The same for B images of ONE shape in one launch (a decoded batch: image i at src_hwc + i * src_image_bytes -> slot i of the
 * [B][3][H][W] tensor; one geometry for all).  With new size == source size this is BasePredictor.preprocess alone: uint8 HWC BGR ->
 * RGB CHW /255 on the device, so a host batch crosses PCIe as 3 bytes per pixel instead of 6 (f16) or 12 (f32). */
int ey_letterbox_batch(int out_dtype, const uint8_t* src_hwc, int B, int src_h, int src_w, int src_row_bytes, long src_image_bytes,
                       void* dst_chw, int H, int W, int new_h, int new_w, int top, int left, int pad_value, int swap_rb,
                       ey_stream_t stream);

/* Linear copy as an ordinary kernel in `stream`: dst (device) <- src (device, or PINNED HOST memory that the device reads over PCIe
 * itself; a small persistent grid then).  The upload path of `YOLO.predict_batches`: on this stack an H2D hipMemcpyAsync does not
 * overlap with kernels of other streams, a kernel does.  nbytes and both pointers 16-byte aligned. */
int ey_copy_linear(const void* src, void* dst, size_t nbytes, ey_stream_t stream);

/* ---- K8a: linear attention core (LinearAttention.forward, block.py:3360-3373) between the qkv and proj convs.
 * qkv [B,N,3C] channel order [q(h0..)|k|v] (block.py:3364); y [B,N,C]:  k=softmax_d(k); q=softmax_N(q);
 * ctx_h = k_h^T v_h; y_h = q_h ctx_h.  head_dim = C/heads <= 64. */
int ey_linear_attention(int dtype, int B, int N, int C, int heads, const void* qkv, int qkv_cstride, void* y,
                        int y_cstride, ey_stream_t stream);

/* ---- K8b: softmax attention core (Attention.forward, block.py:1042-1053) of the YOLO11 baseline PSA block.
 * qkv [B,N,heads*(2*kd+hd)] per-head channel order [q(kd)|k(kd)|v(hd)]; y [B,N,heads*hd] = v @ softmax(q^T k * scale)^T. */
int ey_softmax_attention(int dtype, int B, int N, int heads, int kd, int hd, float scale, const void* qkv,
                         int qkv_cstride, void* y, int y_cstride, ey_stream_t stream);

/* ---- K9+K10: DGQP quality + DFL expectation + anchor decode + score modulation for ONE pyramid level.
 * Detect._inference (head.py:117-148), DFL (block.py:87-90), make_anchors/dist2bbox (utils/tal.py:333-357),
 * GF2Detect._compute_quality_from_logits / _inference_with_quality (head.py:227-243,301-345).
 * box [B,H,W,4*16], cls [B,H,W,nc]; q_* = reg_conf weights (fp32, device) or NULL for plain Detect:
 *   q_w1 [64][20], q_b1 [64], q_w2 [64], q_b2 [1].
 * pred fp32 [B, 4+nc, A_total]; this level fills anchors [a_off, a_off + H*W). */
int ey_head_decode(int dtype, int B, int H, int W, int nc, float stride, const void* box, int box_cstride,
                   const void* cls, int cls_cstride, const float* q_w1, const float* q_b1, const float* q_w2,
                   const float* q_b2, int q_hidden, float* pred, int A_total, int a_off, ey_stream_t stream);

/* All pyramid levels of one head in ONE launch (arrays of length nlevels <= 4; same nc / q_hidden for every level; q_* arrays
 * NULL or holding NULLs for plain Detect).  Same arithmetic as ey_head_decode per level: Detect.forward decodes every level
 * after all towers ran (head.py:84-90,117-148), so the three small launches collapse into one grid that fills the chip. */
int ey_head_decode_levels(int dtype, int B, int nlevels, const int* H, const int* W, const float* stride, const void* const* box,
                          const int* box_cstride, const void* const* cls, const int* cls_cstride, int nc, const float* const* q_w1,
                          const float* const* q_b1, const float* const* q_w2, const float* const* q_b2, int q_hidden, float* pred,
                          int A_total, const int* a_off, ey_stream_t stream);

/* K9+K10 with the NMS candidate build fused in (predict mode, single label; Detect._inference head.py:117-148 followed by the
 * candidate selection of non_max_suppression, utils/ops.py:253,273-275): same arithmetic as ey_head_decode_levels, and per anchor the
 * best class (first maximal index), the sort key (score_bits << 32 | ~anchor) when score > conf_thres and the class passes
 * class_mask (else 0), and the box (cx,cy,w,h) go to `candidates` (ey_nms_candidates_bytes(B, A_total) bytes, 16-byte aligned):
 * exactly what ey_nms derives from `pred`, from the same fp32 values -- so ey_nms_candidates() returns bit-identical rows without the
 * (B,4+nc,A) tensor being written or re-read.  pred_or_null: also write the reference-layout prediction tensor (callers that want it).
 * The levels must cover all A_total anchors. */
size_t ey_nms_candidates_bytes(int B, int A);
int ey_head_decode_levels_nms(int dtype, int B, int nlevels, const int* H, const int* W, const float* stride, const void* const* box, const int* box_cstride,
                              const void* const* cls, const int* cls_cstride, int nc, const float* const* q_w1, const float* const* q_b1,
                              const float* const* q_w2, const float* const* q_b2, int q_hidden, float* pred_or_null, int A_total, const int* a_off,
                              float conf_thres, const uint8_t* class_mask, void* candidates, size_t candidates_bytes, ey_stream_t stream);
/* Second half of ey_nms (selection + greedy suppression, utils/ops.py:277-309 + torchvision.ops.nms) on a candidate buffer
 * written by ey_head_decode_levels_nms.  Outputs as ey_nms.  The tail of the buffer (beyond keys / class ids / boxes) is the scratch of
 * the predict-mode fast path (sorted candidate records + the pairwise suppression bit matrix) and is written by this call. */
int ey_nms_candidates(int B, int nc, int A, void* candidates, size_t candidates_bytes, float iou_thres, int max_det, int max_nms, float max_wh,
                      int agnostic, float* out_boxes, int32_t* out_count, int32_t* out_index, ey_stream_t stream);

/* ---- End2end (NMS-free) heads: E2EDetect = GF2Detect with end2end=True (head.py:273-298,799-824).
 * ey_head_decode_levels_xyxy: ey_head_decode_levels with rows 0-3 of pred = x1,y1,x2,y2 in input pixels (Detect.decode_bboxes passes
 *   xywh = not end2end to dist2bbox, head.py:163-165, utils/tal.py:348-357); rows 4.. unchanged.
 * ey_e2e_topk: Detect.postprocess (head.py:167-189) on that tensor: out_rows [B, k, 6] fp32 = the k best (anchor, class) pairs of each
 *   image in descending score order, row = x1,y1,x2,y2,score,class; k = min(max_det, A) is the caller's; out_index [B, k] (anchor of
 *   each row) or NULL.  Equal scores: lower anchor, then lower class first (torch.topk leaves that order unspecified).
 *   workspace: ey_e2e_topk_workspace_bytes(B, nc, A) bytes, 16-byte aligned. */
int ey_head_decode_levels_xyxy(int dtype, int B, int nlevels, const int* H, const int* W, const float* stride, const void* const* box,
                               const int* box_cstride, const void* const* cls, const int* cls_cstride, int nc, const float* const* q_w1,
                               const float* const* q_b1, const float* const* q_w2, const float* const* q_b2, int q_hidden, float* pred,
                               int A_total, const int* a_off, ey_stream_t stream);
size_t ey_e2e_topk_workspace_bytes(int B, int nc, int A);
int ey_e2e_topk(int B, int nc, int A, const float* pred_xyxy, int k, float* out_rows, int32_t* out_index, void* workspace,
                size_t workspace_bytes, ey_stream_t stream);

/* ---- Block programs: a chain of layers on SMALL feature maps (<= 4096 pixels per image; built for the 20x20 part of the network at
 * 640x640: stride-2 Conv -> DSC3K2_Wavelet -> SPPF -> C2PSA_LinearAttention, nn/modules/block.py:204-223,3412-3497,3749-3788; the
 * last neck block; the 20x20 Detect towers, head.py:59-70) executed by ONE launch: one persistent 1024-thread workgroup per image
 * walks the stages below with a workgroup barrier between them.  Each stage is one of the operators above with the same
 * arithmetic (f16 storage between stages, fp32 accumulate); f16 only.
 *
 * A tensor reference is (addr, ext): ext < 0 -> addr is an absolute device pointer (weights, intermediates owned by the caller for
 * the life of the program); ext >= 0 -> addr is a BYTE offset into the ext-th external tensor handed to ey_block_run (the chain's
 * inputs and outputs, which change from call to call).  *_img = elements between consecutive images of that tensor; *_cs = pixel
 * stride (elements) as everywhere in this ABI. */
enum { EY_BLK_CONV = 0, EY_BLK_DW = 1, EY_BLK_DWT = 2, EY_BLK_POOL = 3, EY_BLK_LINATTN = 4 };
typedef struct {
  int32_t op;
  int32_t H, W, Ho, Wo;        /* input / output extent of the stage */
  int32_t k, stride, act;      /* CONV: k in {1,3}, stride in {1,2}, pad k/2.  DW: k in {3,5,7}, stride 1 */
  int32_t nsrc;                /* CONV: 1 or 2 virtually concatenated sources; others: 1 */
  int64_t src[2]; int32_t src_ext[2]; int64_t src_img[2]; int32_t src_cs[2]; int32_t src_C[2];
  const void* w;               /* CONV: ey_conv_pack_weight layout (group sets w_g elements apart); DW: [k][k][C] f16 */
  const float* bias;           /* CONV: [sets][Cout] fp32 or NULL; DW: [C] fp32 or NULL */
  int64_t w_g; int32_t w_gmax;
  int64_t y; int32_t y_ext; int64_t y_img; int32_t y_cs; int32_t Cout;   /* POOL: y = the y1 slot (y2, y3 follow at C-channel steps); LINATTN: Cout = C */
  int32_t has_res; int64_t res; int32_t res_ext; int64_t res_img; int32_t res_cs;
  int32_t has_addz; int64_t addz; int32_t addz_ext; int64_t addz_img; int32_t addz_cs, addz_H, addz_W;
  float out_scale;
  int32_t ngroup; int64_t src_g, y_g;   /* CONV: channel-offset groups as in ey_conv_desc */
  int32_t heads;               /* LINATTN: heads of 64 channels, even */
  /* filled by ey_block_compile */
  int32_t kpad, nt_pack, mt, nti, lds, tile_nti;
  int32_t tile_src_lds[2], tile_src_lcs[2], tile_y_lds, tile_y_lcs, tile_res_lds, tile_res_lcs, tile_lds_bytes;  /* pointwise-chain LDS placement */
  float zsy, zsx;
} ey_block_stage;
size_t ey_block_stage_sizeof(void);
size_t ey_block_program_bytes(int nstages);
/* host -> host: validate + derive; upload `out_host` (ey_block_program_bytes bytes) to the device afterwards. */
int ey_block_compile(const ey_block_stage* stages_host, int nstages, void* out_host, size_t out_bytes);
/* One launch: grid = B workgroups.  ext_ptrs_host: the `next` (<= 8) external tensors' device base pointers (host array, read at call time). */
int ey_block_run(const void* program_dev, int nstages, int B, const void* const* ext_ptrs_host, int next, ey_stream_t stream);
/* Pointwise chains: when every stage is a 1x1 / stride-1 / single-group conv on one map (ey_block_tileable(compiled program) != 0) the
 * pixels do not interact, and the program runs as one 256-thread workgroup per 32-pixel tile (grid = tiles x B: the whole chip) instead
 * of one workgroup per image, chain-internal tensors in LDS: C2PSA's proj -> ffn -> ffn -> cv2, cv1 -> qkv, DSC3k's cv3 -> cv2 (block.py:3412-3497,1506-1562). */
int ey_block_tileable(const ey_block_stage* compiled_host, int nstages);
int ey_block_run_tiles(const void* program_dev, int nstages, int B, int H, int W, int lds_bytes /* compiled stage 0's tile_lds_bytes */,
                       const void* const* ext_ptrs_host, int next, ey_stream_t stream);
/* Developer tool: the same launch; workgroup 0 also stores wall_clock64() (100 MHz ticks) at the start and after every stage into
 * tstamps_dev[nstages + 1] (tools/block_stage_times.py prints the per-stage split). */
int ey_block_run_timed(const void* program_dev, int nstages, int B, const void* const* ext_ptrs_host, int next, long long* tstamps_dev, ey_stream_t stream);

/* ---- K11: batched per-image NMS (non_max_suppression, utils/ops.py:230-316, and the torchvision.ops.nms it calls
 * at :296).  multi_label=0: best class per anchor (predict, ops.py:273-275); multi_label=1: one candidate per
 * (anchor, class) pair above conf (validation, ops.py:270-272; needs the _ml workspace).
 * pred fp32 [B,4+nc,A] (xywh + scores), NOT modified.
 * out_boxes fp32 [B,max_det,6] = x1,y1,x2,y2,conf,cls in kept (descending score) order; out_count int32 [B];
 * out_index int32 [B,max_det] = anchor index of each kept row (may be NULL); class_mask uint8 [nc] or NULL
 * (the `classes=` filter).  workspace: ey_nms_workspace_bytes(B, A) bytes. */
size_t ey_nms_workspace_bytes(int B, int A);
size_t ey_nms_workspace_bytes_ml(int B, int nc, int A);
int ey_nms(int B, int nc, int A, const float* pred, float conf_thres, float iou_thres, int max_det, int max_nms,
           float max_wh, int agnostic, int multi_label, const uint8_t* class_mask, float* out_boxes, int32_t* out_count,
           int32_t* out_index, void* workspace, size_t workspace_bytes, ey_stream_t stream);

/* ---- Test-time augmentation (DetectionModel._predict_augment, nn/tasks.py:372-408; reached by predict(augment=True)).
 * ey_scale_img = `scale_img(x.flip(3) if flip_lr else x, ratio, gs)` (utils/torch_utils.py:423-432): bilinear resize
 * (align_corners=False, ATen source-index rule) of the planar NCHW image x [B,C,H,W] to hs x ws = int(H*r) x int(W*r), written into the
 * top-left corner of y [B,C,Hp,Wp] (Hp,Wp = ceil(H*r/gs)*gs, ...), the rest filled with pad_value (0.447).  x and y must not alias.
 * ey_tta_merge = `_descale_pred` (:388-397) + the anchor slice `_clip_augmented` (:399-408) keeps + this scale's columns of
 * `torch.cat(y, -1)`: out[b, r, out_off + a - lo] = f(pred[b, r, a]) for a in [lo, hi), rows 0-3 divided by `scale`, row 0 mirrored
 * (img_w - x) when flip == 3, row 1 (img_h - y) when flip == 2; flip in {0, 2, 3}.  pred [B,no,A] and out [B,no,A_out] fp32. */
int ey_scale_img(int dtype, int B, int C, int H, int W, const void* x, int hs, int ws, int Hp, int Wp, int flip_lr, float pad_value, void* y,
                 ey_stream_t stream);
int ey_tta_merge(int B, int no, int A, const float* pred, int lo, int hi, float scale, int flip, int img_h, int img_w, float* out, long A_out, int out_off,
                 ey_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
