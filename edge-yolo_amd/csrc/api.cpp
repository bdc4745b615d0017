// Error reporting and version of libedgeyolo_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/edgeyolo_hip.h"

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
