// What costs the split GEMM loop its MFMA duty?  8 waves/CU, 8 accumulators (128x64 wave tile), per "k-tile" two batches
// of 24 MFMAs, optionally with (R) 12 ds_read_b128 feeding the NEXT batch issued behind the first MFMA of each batch and
// (B) one s_barrier per k-tile.  Prints TFLOP/s of each combination.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool R, bool B>
__global__ __launch_bounds__(512) void loop(float* out, int iters, float scale) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += 512) reinterpret_cast<float*>(lds)[i] = 0.001f * (float)(i & 255);
    __syncthreads();
    const int off = ((wave * 32 + (lane & 31)) & 255) * 128 + ((((lane >> 5) * 2) ^ ((lane >> 1) & 7)) << 4);
    half8 f[2][12];
    for (int h = 0; h < 2; ++h) for (int u = 0; u < 12; ++u) for (int e = 0; e < 8; ++e) f[h][u][e] = (_Float16)(scale * (float)((lane * 37 + u * 11 + e * 5) % 97 - 48));
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (R) {
#pragma unroll
                for (int u = 0; u < 12; ++u) f[h ^ 1][u] = *reinterpret_cast<const half8*>(lds + ((off + u * 4096 + (it * 2 + h) * 16) & 131071));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[h][j], f[h][8 + (j & 3)], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[h][(j + 1) & 7], f[h][8 + (j & 3)], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[h][j], f[h][8 + ((j + 1) & 3)], acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 23, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (B && h == 0) __syncthreads();
        }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += acc[j][0];
    if (s == 123.456f) out[0] = s;
}
template <bool R, bool B>
static void run(float* out, float scale) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)loop<R, B>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop<R, B>), dim3(256), dim3(512), 131072, 0, out, iters, scale);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 8 * iters * 48 * 32768.0;
    printf("operand scale %g ds_reads %d barrier %d: %.1f us, %.0f TFLOP/s (%.1f %% of 2500)\n", scale, (int)R, (int)B, ms * 1e3, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100);
}
int main() {
    float* out; (void)hipMalloc(&out, 16);
    for (float sc : {0.f, 0.01f}) { run<false, false>(out, sc); run<false, true>(out, sc); }
    run<true, true>(out, 0.01f);
    return 0;
}
