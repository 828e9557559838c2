// Error plumbing + version for libserenade_hip.
#include <stdarg.h>
#include <stdio.h>

#include "serenade_hip.h"

static thread_local char g_err[512] = "";

void srn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* srn_last_error(void) { return g_err; }
extern "C" int srn_abi_version(void) { return SRN_ABI_VERSION; }
