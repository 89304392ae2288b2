// error plumbing + version of libcruller_hip.so
#include <cstdarg>
#include <cstdio>
#include "../../include/crl.h"

static thread_local char g_err[512] = "";

extern "C" void crl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* crl_last_error(void) { return g_err; }
extern "C" int crl_version(void) { return 1; }
