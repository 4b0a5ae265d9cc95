#include <stdarg.h>
#include <stdio.h>
#include <hip/hip_runtime.h>
#include "common.h"

static thread_local char g_err[512] = "";

void blt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int blt_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        blt_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return BLT_ERR_HIP;
    }
    return BLT_OK;
}

extern "C" const char* bltvqg_last_error_string(void) { return g_err; }
extern "C" int bltvqg_version(void) { return 100; }
