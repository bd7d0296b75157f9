// Thread-local error text for the C-ABI (include/mreserve_hip.h: mr_last_error).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/mreserve_hip.h"

static thread_local char g_err[512] = "";

void mr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mr_last_error(void) { return g_err; }
extern "C" int mr_version(void) { return 1; }
