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

// x -> (hi, lo) with hi = f16(x), lo = f16(x - hi): the split-f16 pair.  The value is pinned in a register first: without
// that the compiler may fuse the conversion with the FMA that produced x (v_fma_mixlo_f16: ONE rounding from the exact
// product) for the hi half and subtract the separately rounded f32 for the lo half; where x sits exactly between two f16
// values the two disagree on hi and the stored lo gets the wrong sign - an error of 2^-12 |x| on ~1 element in 50 000
// (found by tests/test_gpu_fast.py::test_split_attention_kernel_vs_float64 after an unrelated change to the store pattern).
__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    asm volatile("" : "+v"(x));  // opaque from here on: both halves derive from this one f32
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// A 64-lane xor butterfly on the vector ALU.  __shfl_xor is a ds_bpermute round trip through the LDS crossbar (~100+ cycles of latency
// each; six dependent ones per wave_sum).  Every step below exchanges with exactly lane ^ o, so a sum / max taken with them in the same
// order (32, 16, 8, 4, 2, 1) has the bits of the __shfl_xor form (tools/micro/wave_reduce_check.hip compares every lane of every step):
//   o = 32, 16: v_permlane32_swap / v_permlane16_swap exchange the upper 32 lanes (the odd 16-lane rows) of one register with the lower
//               32 lanes (the even rows) of another - fed two copies of v they leave {v[l & ~o], v[l | o]} in the two.  Inline asm: hipcc
//               7.2 has folded the two results of the builtin called on (u, u) into ONE register inside larger kernels; s_nop 1 on both
//               sides: a VALU result needs two wait states before a permlane reads it and the hazard recogniser does not look inside asm;
//   o = 8:      DPP row_ror:8 (a rotation by half a row IS the xor);
//   o = 4:      DPP row_ror:4 gives lane l the value of lane l - 4 (mod 16): the partner for the quads with bit 2 set (bank mask 0b1010),
//               row_ror:12 the value of lane l + 4 for the others (0b0101);
//   o = 2, 1:   DPP quad_perm [2,3,0,1] / [1,0,3,2].
#define SOLA_DPP_MOV(old, v, ctrl, bank) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (old)), __builtin_bit_cast(int, (v)), (ctrl), 0xF, (bank), false))
__device__ __forceinline__ void swap32_pair(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16_pair(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xor8_of(float v) { return SOLA_DPP_MOV(0.f, v, 0x128, 0xF); }
__device__ __forceinline__ float xor4_of(float v) { return SOLA_DPP_MOV(SOLA_DPP_MOV(0.f, v, 0x124, 0xA), v, 0x12C, 0x5); }
__device__ __forceinline__ float xor2_of(float v) { return SOLA_DPP_MOV(0.f, v, 0x4E, 0xF); }
__device__ __forceinline__ float xor1_of(float v) { return SOLA_DPP_MOV(0.f, v, 0xB1, 0xF); }
__device__ __forceinline__ float sum_xor32(float x) { float a, b; swap32_pair(x, a, b); return a + b; }
__device__ __forceinline__ float sum_xor16(float x) { float a, b; swap16_pair(x, a, b); return a + b; }
__device__ __forceinline__ float max_xor32(float x) { float a, b; swap32_pair(x, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float max_xor16(float x) { float a, b; swap16_pair(x, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float sum_xor8(float x) { return x + xor8_of(x); }
__device__ __forceinline__ float sum_xor4(float x) { return x + xor4_of(x); }
__device__ __forceinline__ float sum_xor2(float x) { return x + xor2_of(x); }
__device__ __forceinline__ float sum_xor1(float x) { return x + xor1_of(x); }

__device__ __forceinline__ float wave_sum(float v) {
    return sum_xor1(sum_xor2(sum_xor4(sum_xor8(sum_xor16(sum_xor32(v))))));
}
__device__ __forceinline__ float wave_max(float v) {
    v = max_xor16(max_xor32(v));
    v = fmaxf(v, xor8_of(v));
    v = fmaxf(v, xor4_of(v));
    v = fmaxf(v, xor2_of(v));
    return fmaxf(v, xor1_of(v));
}
// Sum over the 32-lane half of the wave a lane belongs to, the same value in all of its lanes.  DPP adds inside the 16-lane rows
// (swap pairs, swap quad halves, mirror the 8-lane halves, mirror the row: after each step the sum is uniform over twice as many
// lanes and both partners add the same two numbers, so every lane ends with the bit-identical result), then one
// v_permlane16_swap across the row pair.  VALU only: __shfl_xor would be five ds_bpermute_b32 through the LDS crossbar.
#define SOLA_DPP_ADD(v, ctrl) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, false)))
__device__ __forceinline__ float half_sum32(float v) {
    v = SOLA_DPP_ADD(v, 0xB1);   // quad_perm [1,0,3,2]
    v = SOLA_DPP_ADD(v, 0x4E);   // quad_perm [2,3,0,1]
    v = SOLA_DPP_ADD(v, 0x141);  // row_half_mirror
    v = SOLA_DPP_ADD(v, 0x140);  // row_mirror
    // the swap exchanges the odd rows of its first operand with the even rows of its second: given two copies of v it leaves
    // (row 0, row 0, row 2, row 2) and (row 1, row 1, row 3, row 3).  Inline asm: inside these kernels hipcc 7.2 folds the two
    // results of __builtin_amdgcn_permlane16_swap(u, u) into ONE register (v_add v, v, v: twice the first row pair's partner-
    // less sum; an opaque copy of the operand does not stop it).  s_nop 1: a VALU result needs two wait states before a
    // permlane reads it, and the hazard recogniser does not look inside asm.
    float a = v, b = v;
    // ... and a trailing s_nop 1: no wait-state rule is published for a VALU reading a swap's results, the compiler pads one state
    // behind an asm statement and cannot see what the statement wrote; two more states cost nothing measurable here
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// Sum over the whole wave without the LDS crossbar: half_sum32, then one v_permlane32_swap across the halves.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = half_sum32(v);
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));  // (lo, lo) and (hi, hi) of two copies
    return a + b;
}

// Sum over a 256-thread block; every thread gets the result. `red` is >= 4 floats of LDS; deterministic order.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();  // protect `red` from a previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Counter-based dropout mask (nn.Dropout at module/module.py:78-94, SDPA dropout at tools/attention.py:71): the keep
// decision of element `idx` is a pure function of (seed, idx), so the backward kernels regenerate the forward's mask
// instead of storing it.  keep_thr = (1 - p) * 2^32; p = 0 -> keep_thr = 0xFFFFFFFF and `enabled` is false.
struct DropoutCfg {
    unsigned int seed_lo, seed_hi, keep_thr;
    float scale;  // 1 / (1 - p)
    int enabled;
};
__host__ __device__ __forceinline__ bool dropout_keep(const DropoutCfg& d, unsigned long long idx) {
    unsigned int h = d.seed_lo ^ ((unsigned int)idx * 0x9E3779B1u);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h ^= d.seed_hi + (unsigned int)(idx >> 32) * 0x27D4EB2Fu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h < d.keep_thr;
}
static inline DropoutCfg make_dropout(float p, unsigned long long seed, unsigned int stream_id) {
    DropoutCfg d;
    d.enabled = p > 0.f ? 1 : 0;
    d.seed_lo = (unsigned int)seed ^ (stream_id * 0x9E3779B9u);
    d.seed_hi = (unsigned int)(seed >> 32) + stream_id * 0x7F4A7C15u;
    const double keep = 1.0 - (double)p;
    d.keep_thr = keep >= 1.0 ? 0xFFFFFFFFu : (unsigned int)(keep * 4294967296.0);
    d.scale = (float)(1.0 / keep);
    return d;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Function attributes (MaxDynamicSharedMemorySize) and the CU count belong to a DEVICE: a launcher keeps one
// DeviceOnce per kernel instantiation and runs its set-up the first time each device launches it.
// `if (once.needed(&dev)) { ...; once.done(dev); }` - setting an attribute twice from two racing threads is harmless.
struct DeviceOnce {
    unsigned long long mask = 0;  // bit d: device d has been set up (read/written with relaxed atomics)
    bool needed(int* dev) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) d = 0;
        *dev = d;
        return ((__atomic_load_n(&mask, __ATOMIC_ACQUIRE) >> d) & 1ull) == 0;
    }
    void done(int dev) { __atomic_fetch_or(&mask, 1ull << dev, __ATOMIC_RELEASE); }
};
// multiProcessorCount of the current device (cached per device)
static inline int sola_cu_count() {
    static int n_cu[64] = {0};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) d = 0;
    int v = __atomic_load_n(&n_cu[d], __ATOMIC_RELAXED);
    if (v > 0) return v;
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    __atomic_store_n(&n_cu[d], v, __ATOMIC_RELAXED);
    return v;
}

// Launch check used after every kernel launch.
#define SOLA_LAUNCH_CHECK()                                                                      \
    do {                                                                                         \
        hipError_t _e = hipGetLastError();                                                       \
        if (_e != hipSuccess) {                                                                  \
            sola_set_error("%s:%d: kernel launch failed: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return SOLA_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)
