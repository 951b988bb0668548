// Error reporting, version and device probe for libhwg_hip.so.
#include "hwg_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void hwg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* hwg_last_error(void) { return g_err; }
extern "C" int hwg_abi_version(void) { return 1; }
extern "C" int hwg_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n > 0 ? 1 : 0;
}
