// Error reporting and device probe for the C-ABI (include/yolo_hip.h).
#include "common.hpp"
#include <cstring>

namespace yolo {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace yolo

extern "C" const char* yolo_last_error(void) { return yolo::g_err; }

extern "C" int yolo_abi_version(void) { return 1; }

extern "C" int yolo_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n > 0 ? 1 : 0;
}
