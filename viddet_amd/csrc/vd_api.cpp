// vd_api.cpp — error reporting for the C-ABI (include/viddet_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/viddet_hip.h"

static thread_local char g_err[512] = "";

void vd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
const char* vd_last_error(void) { return g_err; }
int vd_version(void) { return 100; }
}
