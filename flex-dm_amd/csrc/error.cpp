// Thread-local error text for the C-ABI (mfp_last_error).
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

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

// ----------------------------------------------------------------------------- CU count of the current device
// One cache for every launcher (per device: a process may drive several).  mfp_ncu_physical(): what the device has --
// decisions like "is there one document per CU".  mfp_ncu_launch(): what a PERSISTENT launch (one workgroup per CU:
// grouped weight gradients, the weight-stationary products, the single-pass attention backward) sizes its grid for
// = physical - mfp_set_reserved_cus(n): with N > 1 ranks the reserved CUs are left to RCCL's workgroups, which cannot
// share a CU with a 160 KB / 8-wave workgroup.
static int g_reserved_cus = 0;
int mfp_ncu_physical() {
  static int ncu_of[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int& ncu = ncu_of[dev];
  if (ncu == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    ncu = n;
  }
  return ncu;
}
int mfp_ncu_launch() {
  const int n = mfp_ncu_physical() - g_reserved_cus;
  return n < 8 ? 8 : n;
}
extern "C" int mfp_cu_count(void) { return mfp_ncu_physical(); }
extern "C" int mfp_set_reserved_cus(int n) {
  if (n < 0 || n > mfp_ncu_physical() - 8) {
    mfp_set_error("mfp_set_reserved_cus: %d outside [0, #CUs - 8]", n);
    return MFP_EINVAL;
  }
  g_reserved_cus = n;
  return MFP_OK;
}
