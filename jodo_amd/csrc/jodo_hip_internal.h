// Internal helpers shared by the host-side translation units of libjodo_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#ifdef __cplusplus
extern "C" {
#endif
// records the message for jodo_last_error(); returns the (negative) error code passed in
int jodo_set_error(int code, const char* fmt, ...);
// hipGetLastError() after a launch -> 0 or JODO_ERR_LAUNCH (message recorded)
int jodo_check_launch(const char* what);
#ifdef __cplusplus
}
#endif
