// Dispatch tunables of libedgeyolo_hip.so.  Defaults are the measured optimum on MI355X; nothing reads the environment.
// Developer tools change them through the C ABI (ey_tune_set / ey_tune_get, include/edgeyolo_hip.h); they select between
// kernels that compute the same result, never the arithmetic itself.
#pragma once
struct EyTune {
  long tiles_per_wave = 0;      // ws: minimum tiles per wave before adding workgroups (0 = fill all slots)
  long mt2_min_m = 300000;      // ws: 2 pixel blocks per wave from this many output pixels
  long small_m = 100000;        // small-M kernel below this many output pixels ...
  long small_wmb = 48;          // ... while (#tiles x weight bytes) stays below this many MiB
  long halo_min_c = 48;         // 3x3 halo kernel for Cin in [this, 64]
  long ws_lds_kb = 76;          // ws: preferred LDS per workgroup in KiB (2 workgroups per CU)
  long ws_wg_cu = 2;            // ws: workgroups per CU when LDS allows
  long ws_k3_minnt = 0;         // 3x3: use the K-chunked kernel when the weight-stationary tile would cover fewer than this many 16-channel blocks (and not all of Cout)
  long tile_wlds = 1;           // tile kernel weights through LDS: 0 never, 1 always, 2 for stride 1 only
  long tile_s2_minc = 64;       // tile kernel for stride 2 only from this many input channels ...
  long tile_s2_minm = 40000;    // ... and this many output pixels (measured: below, the weight-stationary / halo kernels win)
  long grid_div = 1;            // persistent kernels: launch 1/grid_div of the resident slots
  long c3p = 1;                 // persistent weight-resident 3x3 tile kernel for Cin = 64, stride 1 (0 = off, 2 = every shape it takes)
  long c3p_min_m = 40000;       // ... from this many output pixels
  long c3p_fast = 1;            // ... deferred + interleaved bias/SiLU epilogue where the conv allows it (0 = generic epilogue after the K loop)
  long c3s = 1;                 // 3x3 stream kernel (weights LDS-resident, pixels straight from L2) for Cin in {64, 128, 256} (0 = off)
  long c3s_mt4_m = 100000;      // ... 4 pixel blocks per wave from this many output pixels (else 2)
  long c3s_min_work = 20000;    // ... only for stride-2 convs with at least this many (output pixels x channel tiles) (20000 takes layer 20, 128 -> 128 at 40x40: isolated +0.3 %, pipelined step -0.9 %: less input traffic than the weight-stationary kernel's 8 x 9 re-reads); c3s = 2: everywhere
  long c3s_cfg = 0;             // ... developer knob: force MT * 10 + ring depth (43, 23); 0 = the rule above
  long c3r = 1;                 // register-stationary 3x3 kernel for Cin == 16 (0 = off)
  long tile_minwg = 400;        // stride-1 tile kernel: halve the channel tile while fewer workgroups than this would be launched
  long tile_flat = 1;           // stride-1 tile kernel: flattened tiles fitted to the map (0 = fixed 8 x 32)
  long tile_mink = 0;           // 3x3 tile kernel for K = 9*Cin >= this (huge value = off)
  long pwr_m = 110000;          // register-stationary pointwise kernel from this many output pixels (huge value = off)
  long pwr_frags = 24;          // pwr: at most this many weight fragments (k-steps x 16-channel blocks) per wave
  long pw_m = 110000;           // lean pointwise kernel below this many output pixels (0 = off)
  long pw_waves = 3072;         // pw: prefer the widest channel tile that still leaves this many waves
  long pw_wmb = 64;             // pw: ... while (#16-pixel tiles x weight bytes), the L2->CU weight traffic, stays below this many MiB
  long xcd_map = 31;            // XCD-contiguous work order (ey_xcd_block): bit 0 = depthwise 3x3 strips, 1 = Toeplitz DSConv tiles, 2 = wavelet_z tiles, 3 = 3x3 stream conv, 4 = 3x3 tile conv
  long pwc = 1;                 // enhancer tail conv + the 1x1 behind it as one launch (conv_pwc_kernel; 0 = two launches, 2 = with an agent-scope acquire)
  long pwn = 1;                 // N-split pointwise kernel (small maps, K 128..512, Cout % 128 == 0; 0 = off)
  long pwn_max_m = 110000;      // ... up to this many output pixels
  long pwn_ntw = 0;             // ... developer knob: 16-channel blocks per wave (1, 2; 0 = rule)
  long ds_p = 0;                // dsconv strip kernel: force the strip length (0 = measured rule)
  long tz_kmask = 168;          // Toeplitz dsconv kernel: bit k set = use it for kernel size k (168 = k 3, 5, 7)
  long tz_minpx = 100000;       // ... k = 3/5 only on maps with at least this many pixels (k = 7 always)
  long ds_strip = 1;            // dsconv register-strip kernel (0 = off)
  long dsb_pair = 1;            // DSBottleneck pair (k3 -> k5/k7 DSConv + residual) as one band kernel on small maps (0 = two launches)
  long dsb_max_px = 300000;     // ... for maps up to this many pixels (B x H x W) -- 300 k also takes the C32 k7 pairs at 80x80, batch 32: isolated the band kernel is 10 % slower there than the two Toeplitz launches (44 vs 40 us), in the batch pipeline the step is 1.3 % faster (one launch, no 13 MB round trip; single graph +0.7 %)
  long dsb_rb = 0;              // ... developer knob: rows per band (0 = cost rule)
  long dsb_p2 = 0;              // ... developer knob: strip length of the second stage (1, 2, 4; 0 = rule)
  long dsb_fixed = 100;         // ... cost rule: fixed cost of a workgroup in stencil row-taps
  long stem_pair = 1;           // layers 0 + 1 (3 -> 16 -> 32, both 3x3 stride 2) as one kernel (0 = two launches; developer knob: 2 / 3 / 4 / 5 = 8x16 / 4x16 / 4x32 / 8x32 tiles)
  long pw3 = 1;                 // closing 1x1 of a block + the 64 -> 64 stride-2 3x3 behind it as one kernel (0 = two launches)
  long pw3_skew = 8;            // ... start delay of the second half of the grid in units of 1024 clocks (anti-phase the two workgroups of a CU)
  long pw3_min_px = 100000;     // ... on maps with at least this many pixels (B x H x W)
  long stem_mfma = 1;           // MFMA stem kernel (0 = VALU stem)
  long linattn_mfma = 1;        // MFMA linear-attention kernel (0 = fp32 VALU kernel)
  long softattn_mfma = 1;       // MFMA softmax-attention kernel (0 = fp32 VALU kernel)
  long sppf_min_wg = 128;       // SPPF pooling: shrink the channel group until at least this many workgroups are launched
  long sppf_cv = 0;             // ... developer knob: force 8-channel vectors per workgroup (1, 2, 4, 8); 0 = the rule
  long nms_mask_wg = 0;         // nf_mask: workgroups per image (0 = the measured default)
  long nms_mask_k = 1536;       // ... all-pairs bit matrix over the first this-many of them (multiple of 512); later candidates are tested against the kept boxes on the fly
  long nms_fast_k = 2048;       // predict-mode NMS: the three-kernel fast path over the best K candidates per image (<= 2048; 0 = general kernel only)
};
extern EyTune g_ey_tune;
static inline const EyTune& tune() { return g_ey_tune; }
