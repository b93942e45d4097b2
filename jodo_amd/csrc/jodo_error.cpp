// Thread-local last-error string for the C ABI (never abort(), every entry point returns int).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/jodo_hip.h"
#include "jodo_hip_internal.h"

static thread_local char g_err[512] = "";

extern "C" int jodo_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* jodo_last_error(void) { return g_err; }

extern "C" int jodo_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return jodo_set_error(JODO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return JODO_OK;
}
