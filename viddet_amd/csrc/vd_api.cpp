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
int vd_abi_version(void) { return VD_ABI_VERSION; }
int64_t vd_sizeof_desc(int which) {
    switch (which) {
        case 0: return (int64_t)sizeof(vd_conv_desc);
        case 1: return (int64_t)sizeof(vd_wgrad_desc);
        case 2: return (int64_t)sizeof(vd_head_desc);
        default: return -1;
    }
}
}
