// Error reporting and version of libedgeyolo_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stddef.h>
#include <string.h>
#include "../../include/edgeyolo_hip.h"
#include "tune.h"

static thread_local char g_err[512] = "";

int ey_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* ey_last_error(void) { return g_err; }
extern "C" int ey_version(void) { return 1; }
extern "C" size_t ey_abi_sizeof(int which) { return which == 0 ? sizeof(ey_conv_desc) : which == 1 ? sizeof(ey_conv_direct_desc) : 0; }

EyTune g_ey_tune;
#define EY_T(f) {#f, offsetof(EyTune, f)}
static const struct { const char* name; size_t off; } k_tunables[] = {
    EY_T(tiles_per_wave), EY_T(mt2_min_m), EY_T(small_m), EY_T(small_wmb), EY_T(halo_min_c), EY_T(ws_lds_kb), EY_T(ws_wg_cu), EY_T(ws_k3_minnt), EY_T(tile_wlds),
    EY_T(tile_s2_minc), EY_T(tile_s2_minm), EY_T(grid_div), EY_T(c3r), EY_T(tile_minwg), EY_T(tile_flat), EY_T(tile_mink), EY_T(pwr_m), EY_T(pwr_frags), EY_T(pw_m),
    EY_T(pw_waves), EY_T(pw_wmb), EY_T(ds_p), EY_T(tz_kmask), EY_T(tz_minpx), EY_T(ds_strip), EY_T(stem_mfma), EY_T(linattn_mfma), EY_T(softattn_mfma), EY_T(c3p), EY_T(c3p_min_m), EY_T(c3p_fast), EY_T(c3s), EY_T(c3s_mt4_m), EY_T(c3s_min_work), EY_T(c3s_cfg), EY_T(nms_fast_k), EY_T(nms_mask_wg), EY_T(sppf_min_wg), EY_T(sppf_cv), EY_T(dsb_pair), EY_T(dsb_max_px), EY_T(dsb_rb), EY_T(dsb_fixed), EY_T(dsb_p2), EY_T(stem_pair), EY_T(pw3), EY_T(pw3_min_px), EY_T(pw3_skew), EY_T(pwn), EY_T(pwn_max_m), EY_T(pwn_ntw), EY_T(pwc), EY_T(xcd_map), EY_T(nms_mask_k)};
static long* tunable(const char* name) {
  if (!name) return nullptr;
  for (const auto& t : k_tunables)
    if (!strcmp(t.name, name)) return (long*)((char*)&g_ey_tune + t.off);
  return nullptr;
}
extern "C" int ey_tune_set(const char* name, long value) {
  long* p = tunable(name);
  if (!p) return ey_set_error(EY_EINVAL, "ey_tune_set: unknown tunable '%s'", name ? name : "(null)");
  *p = value;
  return EY_OK;
}
extern "C" long ey_tune_get(const char* name) {
  const long* p = tunable(name);
  return p ? *p : -1;
}
