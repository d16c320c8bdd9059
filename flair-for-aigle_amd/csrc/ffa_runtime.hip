// Error reporting and library identification for libflairhip.
// Contract (include/flairhip.h): every entry point returns 0 on success, a negative FFA_ERR_* code
// for argument / support errors, or a positive hipError_t for launch failures; nothing throws across
// the C boundary; the text of the last failure on this thread is available from ffa_last_error().
#include "ffa_common.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local char g_last_error[512] = "";

void ffa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

int ffa_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ffa_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return FFA_OK;
}

extern "C" const char* ffa_last_error(void) { return g_last_error; }

extern "C" int ffa_version(void) { return 100; }

extern "C" const char* ffa_target_arch(void) { return "gfx950"; }
