// Thread-local error text for the C-ABI (mfp_last_error).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mfp_hip.h"

static thread_local char g_err[512] = "";

void mfp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mfp_last_error(void) { return g_err; }
extern "C" int mfp_version(void) { return 1; }
