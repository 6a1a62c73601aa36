// f32 -> "split-f16" conversion for the 3 x f16-MFMA GEMM path (gemm.hip, ARITH 1).
// Every aligned group of 8 consecutive f32 values x[0..7] of a row becomes, in the SAME 32 bytes,
//   [ f16 hi[0..7] | f16 lo[0..7] ],  hi = f16(s*x), lo = f16(s*x - hi)         (s = a power of two, exact)
// so hi + lo carries 22 significant bits of s*x.  The weight matrices are pre-scaled (s = 64: |w| <= 1/32 would put
// the lo halves in the f16 subnormal range) and the GEMM epilogue multiplies the result by 1/s.
// HBM-bound streaming kernel: one thread per 8-element block (two 16-byte loads, two 16-byte stores).
#include <algorithm>

#include "kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// side (optional): the hi halves once more as plain f16 rows [rows][8 * blocks_per_row] - the plain f16 cast of the same values (the
// split-f16 training forward leaves it for the backward's weight-gradient products: SolaCtx::x16)
__global__ __launch_bounds__(256) void cast_sp16_kernel(const float* __restrict__ in, float* __restrict__ out, long long rows,
                                                        int blocks_per_row, int ld_in, int ld_out, float scale, _Float16* __restrict__ side) {
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
        _Float16 h1, l1;
        split_f16(v[j], h1, l1);
        hi[j] = h1; lo[j] = l1;
    }
    half8* o = reinterpret_cast<half8*>(out + r * ld_out + b * 8);
    o[0] = hi;
    o[1] = lo;
    if (side) *reinterpret_cast<half8*>(side + (r * blocks_per_row + b) * 8) = hi;
}

// block-wide max -> ONE atomic per block (a launch over 0.5 GB has 65 K blocks: an atomic per wave queued 260 K updates on one
// address and took as long as the read itself)
__device__ __forceinline__ void block_amax_commit(float m, unsigned* out) {
    __shared__ float bmax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) bmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(bmax[0], bmax[1]), fmaxf(bmax[2], bmax[3]));
        if (m > 0.f && m <= 3.0e38f) atomicMax(out, __float_as_uint(m));
    }
}

// max |x| of a strided matrix as float bits (non-negative floats order like their bit patterns).  A thread owns one float4
// column of a 64-row slab: coalesced row reads, no index arithmetic in the loop, one atomic per wave.
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ in, unsigned* __restrict__ out, long long rows, int f4_per_row, int ld_in) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.f;
    if (c < f4_per_row) {
        const long long r0 = (long long)blockIdx.y * 64;
        const int nr = (int)min((long long)64, rows - r0);
        const float* p = in + r0 * ld_in + (long long)c * 4;
#pragma unroll 8
        for (int r = 0; r < nr; ++r, p += ld_in) {
            const float4 v = *reinterpret_cast<const float4*>(p);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    }
    block_amax_commit(m, out);
}

// amax_kernel that also leaves the column sums of every 64-row slab in part[slab][cols] (bias gradients: the second, tiny
// pass over the slabs is colsum_kernel's) - the gradient matrix is read once for both.
__global__ __launch_bounds__(256) void amax_colsum_kernel(const float* __restrict__ in, unsigned* __restrict__ out, float* __restrict__ part,
                                                          long long rows, int f4_per_row, int ld_in) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.f;
    if (c < f4_per_row) {
        const long long r0 = (long long)blockIdx.y * 64;
        const int nr = (int)min((long long)64, rows - r0);
        const float* p = in + r0 * ld_in + (long long)c * 4;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int r = 0; r < nr; ++r, p += ld_in) {
            const float4 v = *reinterpret_cast<const float4*>(p);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        *reinterpret_cast<float4*>(part + ((long long)blockIdx.y * f4_per_row + c) * 4) = sum;
    }
    block_amax_commit(m, out);
}

// bf16 storage mode (round 5): bfloat16 has f32's exponent range, so the gradient cast needs no data-dependent scale - and no max|x|
// pass in front of it.  ONE read of the gradient matrix leaves its row-major bf16 cast (the operand of the dW and dX GEMMs) and
// the 64-row slab column sums amax_colsum_kernel leaves (same slabs, same order: the bias gradients keep their bits).  A wave owns
// one slab x 64 float4 columns; the scale slot reads {2^13, 1}: auto_scale() of it is 1 for every later consumer.
typedef __bf16 bf16x4c __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void cast_bf16_colsum_kernel(const float* __restrict__ in, __bf16* __restrict__ out, float* __restrict__ scal,
                                                               float* __restrict__ part, long long rows, int f4_per_row, int ld_in, int ld_out) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const long long slab = (long long)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        scal[0] = 8192.f;
        scal[1] = 1.f;
    }
    const long long r0 = slab * 64;
    if (c >= f4_per_row || r0 >= rows) return;
    const int nr = (int)min((long long)64, rows - r0);
    const float* p = in + r0 * ld_in + (long long)c * 4;
    __bf16* o = out + r0 * ld_out + (long long)c * 4;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int r = 0; r < nr; ++r, p += ld_in, o += ld_out) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        bf16x4c b;
        b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4c*>(o) = b;
    }
    *reinterpret_cast<float4*>(part + (slab * f4_per_row + c) * 4) = sum;
}

// Round 6 (bf16 storage of the step's gradients): the producer - the attention backward, a GroupNorm backward - has already written the
// bfloat16 matrix; what is left of the statistics pass is the 64-row slab column sums (the bias gradients), read from the 2-byte rows: a
// third of cast_bf16_colsum_kernel's bytes.  Same slabs and slot as that kernel; a lane owns eight columns (16 bytes per row).
__global__ __launch_bounds__(256) void colsum_slabs_bf16_kernel(const unsigned short* __restrict__ in, float* __restrict__ scal, float* __restrict__ part,
                                                                long long rows, int c8_per_row, int ld_in) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const long long slab = (long long)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        scal[0] = 8192.f;
        scal[1] = 1.f;
    }
    const long long r0 = slab * 64;
    if (c >= c8_per_row || r0 >= rows) return;
    const int nr = (int)min((long long)64, rows - r0);
    const unsigned short* p = in + r0 * ld_in + (long long)c * 8;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int r = 0; r < nr; ++r, p += ld_in) {
        const uint4 w = *reinterpret_cast<const uint4*>(p);
        const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sum[2 * j] += __builtin_bit_cast(float, u[j] << 16);
            sum[2 * j + 1] += __builtin_bit_cast(float, u[j] & 0xffff0000u);
        }
    }
    float* o = part + (slab * c8_per_row + c) * 8;
    *reinterpret_cast<float4*>(o) = make_float4(sum[0], sum[1], sum[2], sum[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(sum[4], sum[5], sum[6], sum[7]);
}

// scale = 2^(13 - floor(log2(amax))): the largest magnitude lands in [2^13, 2^14), far from the f16 overflow at 65504
__device__ __forceinline__ float auto_scale(unsigned amax_bits, int target = 13) {
    const int e = (int)(amax_bits >> 23) - 127;  // floor(log2(amax)) for normal floats
    if (amax_bits == 0u) return 1.f;
    const int k = min(max(target - e, -100), 100);
    return __uint_as_float((unsigned)(k + 127) << 23);
}

__global__ __launch_bounds__(256) void cast_sp16_auto_kernel(const float* __restrict__ in, float* __restrict__ out, long long rows,
                                                             int blocks_per_row, int ld_in, int ld_out, float* __restrict__ scal) {
    const float scale = auto_scale(reinterpret_cast<const unsigned*>(scal)[0]);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) scal[1] = 1.f / scale;
    if (i >= rows * blocks_per_row) return;
    const long long r = i / blocks_per_row;
    const int b = (int)(i - r * blocks_per_row);
    const float4 v0 = *reinterpret_cast<const float4*>(in + r * ld_in + b * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(in + r * ld_in + b * 8 + 4);
    const float v[8] = {v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale, v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale};
    half8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 h1, l1;
        split_f16(v[j], h1, l1);
        hi[j] = h1; lo[j] = l1;
    }
    half8* o = reinterpret_cast<half8*>(out + r * ld_out + b * 8);
    o[0] = hi;
    o[1] = lo;
}

// ---- the same pair of passes for a set of equally shaped matrices (the 12 * n_layers projection weights): blockIdx.z picks
//      the matrix, its scale pair sits at scal + 2 * z
constexpr int MULTI_MAX = 64;
struct MultiArgs {
    const float* in[MULTI_MAX];
    float* out[MULTI_MAX];
};
__global__ __launch_bounds__(256) void amax_multi_kernel(const MultiArgs a, unsigned* __restrict__ scal, long long n4) {
    const float* in = a.in[blockIdx.z];
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = *reinterpret_cast<const float4*>(in + i * 4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    block_amax_commit(m, scal + 2 * blockIdx.z);
}
__global__ __launch_bounds__(256) void cast_sp16_auto_multi_kernel(const MultiArgs a, float* __restrict__ scal, long long n8) {
    const float* in = a.in[blockIdx.z];
    float* out = a.out[blockIdx.z];
    float* sc = scal + 2 * blockIdx.z;
    const float scale = auto_scale(reinterpret_cast<const unsigned*>(sc)[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) sc[1] = 1.f / scale;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const float4 v0 = *reinterpret_cast<const float4*>(in + i * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(in + i * 8 + 4);
        const float v[8] = {v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale, v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale};
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 h1, l1;
            split_f16(v[j], h1, l1);
            hi[j] = h1; lo[j] = l1;
        }
        half8* o = reinterpret_cast<half8*>(out + i * 8);
        o[0] = hi;
        o[1] = lo;
    }
}

// ---- f32 -> plain f16 rows (the 16-bit storage mode): 8 values per thread, 16-byte stores --------------------------------
// SCALED: the power-of-two scale comes from the amax slot (cast_sp16_auto_kernel's rule), else from the argument
// eight scaled f32 values -> eight 16-bit values: f16, or (bf) bfloat16 - both round to nearest even; the bits travel as a half8
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
__device__ __forceinline__ half8 cvt16x8(const float4 v0, const float4 v1, float scale, int bf) {
    const float x[8] = {v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale, v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale};
    if (bf) {
        bf16x8c b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = (__bf16)x[j];
        return __builtin_bit_cast(half8, b);
    }
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)x[j];
    return h;
}

template <bool SCALED>
__global__ __launch_bounds__(256) void cast_f16_kernel(const float* __restrict__ in, _Float16* __restrict__ out, long long rows,
                                                       int blocks_per_row, int ld_in, int ld_out, float scale_arg, float* __restrict__ scal,
                                                       int target, float* __restrict__ scale_out, int bf) {
    float scale = scale_arg;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (SCALED) {
        // scale_arg > 0 caps the data-dependent scale (an output kept in the scaled units also carries scale * bias)
        scale = auto_scale(reinterpret_cast<const unsigned*>(scal)[0], target);
        if (scale_arg > 0.f) scale = fminf(scale, scale_arg);
        if (i == 0) {
            scal[1] = 1.f / scale;
            if (scale_out) *scale_out = scale;
        }
    }
    if (i >= rows * blocks_per_row) return;
    const long long r = i / blocks_per_row;
    const int b = (int)(i - r * blocks_per_row);
    const float4 v0 = *reinterpret_cast<const float4*>(in + r * ld_in + b * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(in + r * ld_in + b * 8 + 4);
    *reinterpret_cast<half8*>(out + r * ld_out + b * 8) = cvt16x8(v0, v1, scale, bf);
}
__global__ __launch_bounds__(256) void cast_f16_auto_multi_kernel(const MultiArgs a, float* __restrict__ scal, long long n8, int bf) {
    const float* in = a.in[blockIdx.z];
    _Float16* out = reinterpret_cast<_Float16*>(a.out[blockIdx.z]);
    float* sc = scal + 2 * blockIdx.z;
    const float scale = auto_scale(reinterpret_cast<const unsigned*>(sc)[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) sc[1] = 1.f / scale;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const float4 v0 = *reinterpret_cast<const float4*>(in + i * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(in + i * 8 + 4);
        *reinterpret_cast<half8*>(out + i * 8) = cvt16x8(v0, v1, scale, bf);
    }
}

// one block per GroupNorm: rms over the channels of sqrt(gamma^2 + beta^2) = the rms of that norm's output
constexpr int NORM_MAX = 32;
struct NormArgs {
    NormPair n[NORM_MAX];
};
__global__ __launch_bounds__(256) void norm_range_check_kernel(const NormArgs a, int* guard) {
    __shared__ float red[4];
    const NormPair np = a.n[blockIdx.x];
    float s = 0.f;
    for (int c = threadIdx.x; c < np.C; c += 256) s += np.gamma[c] * np.gamma[c] + np.beta[c] * np.beta[c];
    const float rms = sqrtf(block_sum_256(s, red) / (float)np.C);
    if (threadIdx.x == 0 && !(rms >= 0.015625f && rms <= 512.f)) atomicOr(guard, 2);
}

}  // namespace

int launch_cast_sp16_auto_multi(const float* const* in, float* const* out, int n, int rows, int K, float* scal, hipStream_t s) {
    SOLA_ARG(in && out && scal && n > 0 && rows > 0 && K > 0 && K % 8 == 0, "cast_sp16_auto_multi: n=%d rows=%d K=%d", n, rows, K);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 12.0 * n * rows * K);
    SOLA_HIP(hipMemsetAsync(scal, 0, (size_t)2 * n * sizeof(float), s));
    const long long elems = (long long)rows * K;
    const unsigned blocks = (unsigned)std::min<long long>(128, (elems / 8 + 255) / 256);
    for (int i0 = 0; i0 < n; i0 += MULTI_MAX) {
        const int nn = std::min(MULTI_MAX, n - i0);
        MultiArgs a;
        for (int i = 0; i < nn; ++i) { a.in[i] = in[i0 + i]; a.out[i] = out[i0 + i]; }
        for (int i = nn; i < MULTI_MAX; ++i) { a.in[i] = nullptr; a.out[i] = nullptr; }
        hipLaunchKernelGGL(amax_multi_kernel, dim3(blocks, 1, nn), dim3(256), 0, s, a, reinterpret_cast<unsigned*>(scal + 2 * i0), elems / 4);
        SOLA_LAUNCH_CHECK();
        hipLaunchKernelGGL(cast_sp16_auto_multi_kernel, dim3(blocks, 1, nn), dim3(256), 0, s, a, scal + 2 * i0, elems / 8);
        SOLA_LAUNCH_CHECK();
    }
    return SOLA_OK;
}

// max|x| does not care about the matrix shape: contiguous rows narrower than 1024 values are re-cut into 1024-wide ones (amax_kernel
// gives a thread one float4 column of a 64-row slab, so a 256-wide matrix - the object tokens - kept 64 of a block's 256 threads busy:
// 74 us for 134 MB)
static void amax_shape(long long& rows, int& K, int& ld) {
    if (ld == K && K < 1024 && (rows * K) % 1024 == 0) {
        rows = rows * K / 1024;
        K = ld = 1024;
    }
}

int launch_cast_f16(const float* in, int ld_in, void* out, int ld_out, long long rows, int K, float scale, float* scal, hipStream_t s,
                    int target_exp, float* scale_out, int bf16) {
    SOLA_ARG(in && out && rows > 0 && K > 0 && K % 8 == 0 && ld_in % 4 == 0 && ld_out % 8 == 0, "cast_f16: K=%d ld_in=%d ld_out=%d", K, ld_in, ld_out);
    const long long n = rows * (K / 8);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, (scal ? 10.0 : 6.0) * rows * K);
    if (scal) {
        SOLA_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(float), s));
        // bfloat16 has f32's exponent range: no max|x| pass - the zeroed slot reads as "scale 1" (auto_scale(0)), the same bits as any
        // power-of-two scale would give, one read of the matrix less (round 6)
        if (!bf16) {
            long long ar = rows; int ak = K, ald = ld_in;
            amax_shape(ar, ak, ald);
            hipLaunchKernelGGL(amax_kernel, dim3((unsigned)((ak / 4 + 255) / 256), (unsigned)((ar + 63) / 64)), dim3(256), 0, s, in,
                               reinterpret_cast<unsigned*>(scal), ar, ak / 4, ald);
        }
        SOLA_LAUNCH_CHECK();
        hipLaunchKernelGGL(cast_f16_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, static_cast<_Float16*>(out), rows, K / 8, ld_in, ld_out,
                           scale_out ? scale : 0.f, scal, target_exp, scale_out, bf16);
    } else {
        hipLaunchKernelGGL(cast_f16_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, static_cast<_Float16*>(out), rows, K / 8, ld_in, ld_out, scale, nullptr,
                           0, nullptr, bf16);
    }
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

// the cast half of launch_cast_f16 with a data-dependent scale: scal[0] already holds max|in| (launch_amax_colsum)
int launch_cast_f16_scaled(const float* in, int ld_in, void* out, int ld_out, long long rows, int K, float* scal, hipStream_t s, int bf16) {
    SOLA_ARG(in && out && scal && rows > 0 && K > 0 && K % 8 == 0 && ld_in % 4 == 0 && ld_out % 8 == 0, "cast_f16_scaled: K=%d ld_in=%d ld_out=%d", K, ld_in, ld_out);
    const long long n = rows * (K / 8);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 6.0 * rows * K);
    hipLaunchKernelGGL(cast_f16_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, static_cast<_Float16*>(out), rows, K / 8, ld_in, ld_out,
                       0.f, scal, 13, nullptr, bf16);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_cast_f16_auto_multi(const float* const* in, void* const* out, int n, int rows, int K, float* scal, hipStream_t s, int bf16) {
    SOLA_ARG(in && out && scal && n > 0 && rows > 0 && K > 0 && K % 8 == 0, "cast_f16_auto_multi: n=%d rows=%d K=%d", n, rows, K);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 10.0 * n * rows * K);
    SOLA_HIP(hipMemsetAsync(scal, 0, (size_t)2 * n * sizeof(float), s));
    const long long elems = (long long)rows * K;
    const unsigned blocks = (unsigned)std::min<long long>(1024, (elems / 8 + 255) / 256);
    for (int i0 = 0; i0 < n; i0 += MULTI_MAX) {
        const int nn = std::min(MULTI_MAX, n - i0);
        MultiArgs a;
        for (int i = 0; i < MULTI_MAX; ++i) { a.in[i] = i < nn ? in[i0 + i] : nullptr; a.out[i] = i < nn ? static_cast<float*>(out[i0 + i]) : nullptr; }
        if (!bf16) {  // bfloat16: scale 1 (the zeroed slots), no max|x| pass over the matrices
            hipLaunchKernelGGL(amax_multi_kernel, dim3(blocks, 1, nn), dim3(256), 0, s, a, reinterpret_cast<unsigned*>(scal + 2 * i0), elems / 4);
            SOLA_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(cast_f16_auto_multi_kernel, dim3(blocks, 1, nn), dim3(256), 0, s, a, scal + 2 * i0, elems / 8, bf16);
        SOLA_LAUNCH_CHECK();
    }
    return SOLA_OK;
}

int launch_norm_range_check(const NormPair* norms, int n, int* guard, hipStream_t s) {
    SOLA_ARG(norms && guard && n > 0 && n <= NORM_MAX, "norm_range_check: n=%d (1..%d)", n, NORM_MAX);
    NormArgs a;
    for (int i = 0; i < NORM_MAX; ++i) a.n[i] = norms[i < n ? i : 0];
    hipLaunchKernelGGL(norm_range_check_kernel, dim3(n), dim3(256), 0, s, a, guard);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_amax_accumulate(const float* in, int ld_in, long long rows, int K, float* scal, hipStream_t s) {
    SOLA_ARG(in && scal && rows > 0 && K > 0 && K % 4 == 0 && ld_in % 4 == 0, "amax: K=%d ld_in=%d", K, ld_in);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 4.0 * rows * K);
    amax_shape(rows, K, ld_in);
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)((K / 4 + 255) / 256), (unsigned)((rows + 63) / 64)), dim3(256), 0, s, in,
                       reinterpret_cast<unsigned*>(scal), rows, K / 4, ld_in);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_amax_colsum(const float* in, int ld_in, long long rows, int K, float* scal, float* part, hipStream_t s) {
    SOLA_ARG(in && scal && part && rows > 0 && K > 0 && K % 4 == 0 && ld_in % 4 == 0, "amax_colsum: K=%d ld_in=%d", K, ld_in);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 4.0 * rows * K);
    hipLaunchKernelGGL(amax_colsum_kernel, dim3((unsigned)((K / 4 + 255) / 256), (unsigned)((rows + 63) / 64)), dim3(256), 0, s, in,
                       reinterpret_cast<unsigned*>(scal), part, rows, K / 4, ld_in);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_cast_bf16_colsum(const float* in, int ld_in, void* out, int ld_out, long long rows, int K, float* scal, float* part, hipStream_t s) {
    SOLA_ARG(in && out && scal && part && rows > 0 && K > 0 && K % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && ld_out >= K,
             "cast_bf16_colsum: K=%d ld_in=%d ld_out=%d", K, ld_in, ld_out);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 6.0 * rows * K);
    const long long slabs = (rows + 63) / 64;
    hipLaunchKernelGGL(cast_bf16_colsum_kernel, dim3((unsigned)((K / 4 + 63) / 64), (unsigned)((slabs + 3) / 4)), dim3(256), 0, s, in,
                       reinterpret_cast<__bf16*>(out), scal, part, rows, K / 4, ld_in, ld_out);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_colsum_slabs_bf16(const void* in, int ld_in, long long rows, int K, float* scal, float* part, hipStream_t s) {
    SOLA_ARG(in && scal && part && rows > 0 && K > 0 && K % 8 == 0 && ld_in % 8 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0, "colsum_slabs_bf16: K=%d ld_in=%d", K, ld_in);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 2.0 * rows * K);
    const long long slabs = (rows + 63) / 64;
    hipLaunchKernelGGL(colsum_slabs_bf16_kernel, dim3((unsigned)((K / 8 + 63) / 64), (unsigned)((slabs + 3) / 4)), dim3(256), 0, s,
                       static_cast<const unsigned short*>(in), scal, part, rows, K / 8, ld_in);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_cast_sp16_scaled(const float* in, int ld_in, float* out, int ld_out, long long rows, int K, float* scal, hipStream_t s) {
    SOLA_ARG(in && out && scal && rows > 0 && K > 0 && K % 8 == 0 && ld_in % 4 == 0 && ld_out % 8 == 0, "cast_sp16_scaled: K=%d ld_in=%d ld_out=%d", K, ld_in, ld_out);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 8.0 * rows * K);
    const long long n = rows * (K / 8);
    hipLaunchKernelGGL(cast_sp16_auto_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, rows, K / 8, ld_in, ld_out, scal);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_cast_sp16_auto(const float* in, int ld_in, float* out, int ld_out, long long rows, int K, float* scal, hipStream_t s) {
    SOLA_ARG(in && out && scal && rows > 0 && K > 0 && K % 8 == 0 && ld_in % 4 == 0 && ld_out % 8 == 0, "cast_sp16_auto: K=%d ld_in=%d ld_out=%d", K, ld_in, ld_out);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 12.0 * rows * K);
    SOLA_HIP(hipMemsetAsync(scal, 0, 2 * sizeof(float), s));
    {
        long long ar = rows; int ak = K, ald = ld_in;
        amax_shape(ar, ak, ald);
        hipLaunchKernelGGL(amax_kernel, dim3((unsigned)((ak / 4 + 255) / 256), (unsigned)((ar + 63) / 64)), dim3(256), 0, s, in,
                           reinterpret_cast<unsigned*>(scal), ar, ak / 4, ald);
    }
    SOLA_LAUNCH_CHECK();
    const long long n = rows * (K / 8);
    hipLaunchKernelGGL(cast_sp16_auto_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, rows, K / 8, ld_in, ld_out, scal);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_cast_sp16(const float* in, int ld_in, float* out, int ld_out, long long rows, int K, float scale, hipStream_t s, void* side16) {
    SOLA_ARG(in && out && rows > 0 && K > 0 && K % 8 == 0 && ld_in % 4 == 0 && ld_out % 8 == 0, "cast_sp16: K=%d ld_in=%d ld_out=%d", K, ld_in, ld_out);
    const long long n = rows * (K / 8);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, (side16 ? 10.0 : 8.0) * rows * K);
    hipLaunchKernelGGL(cast_sp16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, rows, K / 8, ld_in, ld_out, scale,
                       static_cast<_Float16*>(side16));
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
