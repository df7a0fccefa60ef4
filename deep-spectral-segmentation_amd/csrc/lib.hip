// lib.hip - library-level entry points of libdss_hip.so (version, error string).
#include <stdarg.h>

#include "common.h"

namespace dss {
char* err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace dss

extern "C" int dss_abi_version(void) { return DSS_ABI_VERSION; }
extern "C" const char* dss_last_error(void) { return dss::err_buf(); }
extern "C" const char* dss_target_arch(void) { return "gfx950"; }
