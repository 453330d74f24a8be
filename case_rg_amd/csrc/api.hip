// Error reporting and version for libcase_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

int case_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int case_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return CASE_OK;
  return case_set_error(CASE_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

extern "C" int case_version(void) { return 100; /* 0.1.0: round 1 */ }
extern "C" const char* case_last_error(void) { return g_err; }
