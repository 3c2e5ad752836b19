// thread-local error string + ABI version for libdanet_hip.so
#include <stdarg.h>
#include <stdio.h>
#include "danet_hip.h"

static thread_local char g_err[512] = "";

extern "C" void danet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* danet_last_error(void) { return g_err; }
extern "C" int danet_abi_version(void) { return DANET_ABI_VERSION; }
