// api.cpp — error plumbing + version/device queries of libprime_mi355x.
#include "common.hip.h"
#include <stdarg.h>
#include <string.h>

namespace pm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace pm

extern "C" int pm_abi_version(void) { return PM_ABI_VERSION; }
extern "C" const char* pm_last_error(void) { return pm::g_err; }
extern "C" int pm_device_arch(char* buf, int buf_len) {
  if (!buf || buf_len <= 0) return PM_ERR_INVALID_ARG;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    pm::set_error("pm_device_arch: no HIP device");
    buf[0] = 0;
    return PM_ERR_LAUNCH;
  }
  strncpy(buf, prop.gcnArchName, buf_len - 1);
  buf[buf_len - 1] = 0;
  return PM_OK;
}
