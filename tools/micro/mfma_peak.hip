// Sustained v_mfma_f32_32x32x16_f16 rate with no memory traffic at all: NACC independent accumulators per wave,
// `waves` waves per CU.  Tells how much of the 2.5 PFLOP/s paper peak the chip holds under a pure MFMA load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) s += acc[j][0];
    if (s == 123.456f) out[0] = s;
}
template <int NACC>
static void run(float* out, int waves) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(256), dim3(waves * 64), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * waves * iters * 3 * NACC * 32768.0;
    printf("waves/CU %d, %d accumulators: %.1f us, %.0f TFLOP/s (%.1f %% of 2500)\n", waves, NACC, ms * 1e3, flops / (ms * 1e-3) / 1e12,
           flops / (ms * 1e-3) / 2.5e15 * 100);
}
int main() {
    float* out; (void)hipMalloc(&out, 16);
    run<4>(out, 4); run<4>(out, 8); run<8>(out, 4); run<8>(out, 8); run<1>(out, 8); run<2>(out, 8);
    return 0;
}
