// Does a larger wave tile buy sustained MFMA rate on this (power-limited) part?  The split GEMM's k-loop step in two shapes, both with
// the next step's fragment reads behind the first MFMA and one barrier per two steps, operands read from LDS (real bit patterns):
//   8 waves/CU  x  8 accumulators (128x64 wave tile): 12 ds_read_b128 per 24 MFMAs  - today's kernel, two waves per SIMD
//   4 waves/CU  x 16 accumulators (128x128 wave tile): 16 ds_read_b128 per 48 MFMAs - one wave per SIMD, 33 % fewer LDS bytes per MFMA
// Prints TFLOP/s of each (~30 ms launches: long enough for the power manager to settle).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// ORDER 1: the same products with the A operand held over consecutive MFMAs (ah_i x {bh_0, bh_1, bl_0, bl_1}, al_i x {bh_0, bh_1}) and no
// two consecutive MFMAs on one accumulator - does operand reuse between neighbours (less toggling) or chain spacing change the rate?
template <int NJ, int ORDER = 0>  // accumulators = 4 x NJ; NJ = 2: 128x64, NJ = 4: 128x128
__global__ __launch_bounds__(NJ == 2 ? 512 : 256) void loop(float* out, int iters) {
    extern __shared__ char lds[];
    constexpr int NT = NJ == 2 ? 512 : 256, NRD = 2 * (4 + NJ);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += NT) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (unsigned)(i * 2654435761u >> 12 & 0x03ff03ffu);
    __syncthreads();
    const int off = ((wave * 32 + (lane & 31)) & 255) * 128 + ((((lane >> 5) * 2) ^ ((lane >> 1) & 7)) << 4);
    half8 f[2][NRD];
    for (int h = 0; h < 2; ++h) for (int u = 0; u < NRD; ++u) f[h][u] = *reinterpret_cast<const half8*>(lds + ((off + u * 4096) & 131071));
    f32x16 acc[4][NJ];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < NJ; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int u = 0; u < NRD; ++u) f[h ^ 1][u] = *reinterpret_cast<const half8*>(lds + ((off + u * 4096 + (it * 2 + h) * 16) & 131071));
            if constexpr (ORDER == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {  // a[i] = (hi f[2i], lo f[2i+1]); b[j] = (hi f[8+2j], lo f[8+2j+1])
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[h][2 * i + 1], f[h][8 + 2 * j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[h][2 * i], f[h][8 + 2 * j + 1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[h][2 * i], f[h][8 + 2 * j], acc[i][j], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[h][2 * i + 1], f[h][8 + 2 * j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[h][2 * i], f[h][8 + 2 * j + 1], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[h][2 * i], f[h][8 + 2 * j], acc[i][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12 * NJ - 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (h == 0) __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < NJ; ++j) s += acc[i][j][0];
    if (s == 123.456f) out[0] = s;
}
template <int NJ, int ORDER = 0>
static void run(float* out, int iters) {
    hipFuncSetAttribute((const void*)loop<NJ, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop<NJ, ORDER>), dim3(256), dim3(NJ == 2 ? 512 : 256), 131072, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = 256.0 * (NJ == 2 ? 8 : 4) * iters * 2.0 * (12 * NJ) * 32768.0;
    printf("order %d: %d waves/CU x %2d accumulators (128x%d), %d reads per %d MFMAs: %.2f ms, %.0f TFLOP/s executed (%.1f %% of 2500)\n", ORDER, NJ == 2 ? 8 : 4, 4 * NJ, 32 * NJ,
           2 * (4 + NJ), 12 * NJ, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100);
}
int main() {
    float* out; (void)hipMalloc(&out, 16);
    for (int rep = 0; rep < 2; ++rep) { run<2>(out, 10000); run<4>(out, 10000); run<2, 1>(out, 10000); run<4, 1>(out, 10000); }
    return 0;
}
