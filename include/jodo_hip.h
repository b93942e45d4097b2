/* libjodo_hip.so — C ABI of the MI355X-native JODO DGT denoising hot path.  (placeholder; grows) */
#ifndef JODO_HIP_H
#define JODO_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define JODO_OK 0
#define JODO_ERR_ARG (-1)
#define JODO_ERR_LAUNCH (-2)
#define JODO_ERR_UNSUPPORTED (-3)
const char* jodo_last_error(void);
int jodo_debug_mlp(const float* x, int rows, const float* w1, const float* b1, const float* w2,
                   const float* b2, float* y, void* stream);
#ifdef __cplusplus
}
#endif
#endif
