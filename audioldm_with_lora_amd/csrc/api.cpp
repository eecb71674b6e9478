// Version / error reporting for the C-ABI (include/aldm_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "aldm_hip.h"

static thread_local char g_err[512] = "";

void aldm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* aldm_last_error(void) { return g_err; }
extern "C" const char* aldm_version(void) { return "aldm_hip 0.1 (gfx950)"; }
