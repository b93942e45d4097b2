// TEST INFRASTRUCTURE ONLY — a minimal sequential stand-in for <hip/hip_runtime.h>.
//
// The training-path kernels of jodo_amd/csrc/train_ops.hip are written cooperation-free (no LDS, no barriers, no cross-lane
// operations: one thread per output element or per reduced row), so every launch is equivalent to running its threads one after
// another.  tests/emul/Makefile compiles THOSE SAME SOURCES with g++ against this header (it shadows the real one on the include
// path; the product sources carry no #ifdef for it) into tests/emul/libjodo_train_emul.so, which the CPU suite drives with host
// pointers and compares with torch.autograd through the oracle: the index arithmetic and the calculus of every training kernel are
// checked in the build container, where there is no GPU.  The MFMA GEMM (train_gemm.hip) is not emulated: emul_gemm.cpp supplies
// plain loops behind the same launcher signature, and the kernel itself is tested on the GPU.
// Nothing under jodo_amd/ or bench.py loads this library.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_idx { unsigned x, y, z; };
extern thread_local emu_idx threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToDevice = 3, hipMemcpyDeviceToHost = 2 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulation"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

template <typename... KArgs, typename... Args>
inline void emu_launch(void (*k)(KArgs...), dim3 g, dim3 b, Args... args) {
    gridDim = g; blockDim = b;
    for (unsigned bz = 0; bz < g.z; ++bz)
        for (unsigned by = 0; by < g.y; ++by)
            for (unsigned bx = 0; bx < g.x; ++bx) {
                blockIdx = {bx, by, bz};
                for (unsigned tz = 0; tz < b.z; ++tz)
                    for (unsigned ty = 0; ty < b.y; ++ty)
                        for (unsigned tx = 0; tx < b.x; ++tx) {
                            threadIdx = {tx, ty, tz};
                            k(static_cast<KArgs>(args)...);
                        }
            }
}
#define hipLaunchKernelGGL(k, g, b, shmem, stream, ...) emu_launch(k, g, b, __VA_ARGS__)
