// Error reporting / version entry points of libcsbsr_hip.so.
#include <cstdarg>
#include <cstdio>
#include "../../include/csbsr_hip.h"

static thread_local char g_err[512] = "";

void csbsr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int csbsr_version(void) { return 100; }
extern "C" const char* csbsr_last_error(void) { return g_err; }
