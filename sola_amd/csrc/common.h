// Shared host/device helpers for libsola_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sola_hip.h"

void sola_set_error(const char* fmt, ...);

#define SOLA_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            sola_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return SOLA_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

#define SOLA_ARG(cond, ...)                   \
    do {                                      \
        if (!(cond)) {                        \
            sola_set_error(__VA_ARGS__);      \
            return SOLA_ERR_ARG;              \
        }                                     \
    } while (0)

#define SOLA_TRY(expr)            \
    do {                          \
        int _s = (expr);          \
        if (_s != SOLA_OK) return _s; \
    } while (0)

// ---- in-library event profiler (profile.cpp) -------------------------------------------------------------------
struct SolaProfScope {
    int cat;
    hipStream_t stream;
    bool on;
    int slot;
    SolaProfScope(int cat, hipStream_t stream, double flops, double bytes);
    ~SolaProfScope();
};

// ---- device helpers --------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Sum over a 256-thread block; every thread gets the result. `red` is >= 4 floats of LDS; deterministic order.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();  // protect `red` from a previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Launch check used after every kernel launch.
#define SOLA_LAUNCH_CHECK()                                                                      \
    do {                                                                                         \
        hipError_t _e = hipGetLastError();                                                       \
        if (_e != hipSuccess) {                                                                  \
            sola_set_error("%s:%d: kernel launch failed: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return SOLA_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)
