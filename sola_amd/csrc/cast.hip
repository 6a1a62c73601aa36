// f32 -> "split-f16" conversion for the 3 x f16-MFMA GEMM path (gemm.hip, ARITH 1).
// Every aligned group of 8 consecutive f32 values x[0..7] of a row becomes, in the SAME 32 bytes,
//   [ f16 hi[0..7] | f16 lo[0..7] ],  hi = f16(s*x), lo = f16(s*x - hi)         (s = a power of two, exact)
// so hi + lo carries 22 significant bits of s*x.  The weight matrices are pre-scaled (s = 64: |w| <= 1/32 would put
// the lo halves in the f16 subnormal range) and the GEMM epilogue multiplies the result by 1/s.
// HBM-bound streaming kernel: one thread per 8-element block (two 16-byte loads, two 16-byte stores).
#include "kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void cast_sp16_kernel(const float* __restrict__ in, float* __restrict__ out, long long rows,
                                                        int blocks_per_row, int ld_in, int ld_out, float scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * blocks_per_row) return;
    const long long r = i / blocks_per_row;
    const int b = (int)(i - r * blocks_per_row);
    const float4 v0 = *reinterpret_cast<const float4*>(in + r * ld_in + b * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(in + r * ld_in + b * 8 + 4);
    const float v[8] = {v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale, v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale};
    half8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (_Float16)v[j];
        lo[j] = (_Float16)(v[j] - (float)hi[j]);
    }
    half8* o = reinterpret_cast<half8*>(out + r * ld_out + b * 8);
    o[0] = hi;
    o[1] = lo;
}

}  // namespace

int launch_cast_sp16(const float* in, int ld_in, float* out, int ld_out, long long rows, int K, float scale, hipStream_t s) {
    SOLA_ARG(in && out && rows > 0 && K > 0 && K % 8 == 0 && ld_in % 4 == 0 && ld_out % 8 == 0, "cast_sp16: K=%d ld_in=%d ld_out=%d", K, ld_in, ld_out);
    const long long n = rows * (K / 8);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 8.0 * rows * K);
    hipLaunchKernelGGL(cast_sp16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, rows, K / 8, ld_in, ld_out, scale);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
