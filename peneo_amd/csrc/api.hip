// Error plumbing + version of libpeneo_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace peneo {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return PENEO_ERR_LAUNCH;
  }
  return PENEO_OK;
}

}  // namespace peneo

extern "C" int peneo_version(void) { return 100; }
extern "C" const char* peneo_last_error(void) { return peneo::g_err; }
