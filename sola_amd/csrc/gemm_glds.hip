// Split-f16 NT GEMM with direct-to-LDS staging (global_load_lds_dwordx4): the staging-bound variant of gemm.hip's
// ARITH 1 kernel.  Same math (three v_mfma_f32_32x32x16_f16 per product on (hi, lo) f16 pairs, f32 accumulate), same
// 128x128x32 block tile and 2x2 wave grid; what changes is how a k-tile reaches LDS:
//   * every wave issues eight 1-KiB LDS-DMA pieces per k-tile (4 for A, 4 for W): no VGPR round trip, no ds_write, no
//     per-element address arithmetic in the loop;
//   * the DMA writes LDS linearly (wave-uniform base + lane * 16 B), so tile rows are 128 contiguous bytes (8 lanes
//     per row, 8 rows per piece) and bank conflicts are removed by an XOR swizzle of the 32-byte blocks,
//     block' = block ^ ((row >> 1) & 3), applied on the SOURCE address when loading and on the fragment address when
//     reading (the same involution on both sides);
//   * rows outside the matrix / conv padding rows read a 16-byte zero page instead (the DMA cannot predicate data).
// Two LDS stages (64 KiB per block, two blocks per CU); the DMA of tile k+1 is in flight during tile k's MFMAs and is
// drained (vmcnt(0)) right before the barrier that publishes it.
#include "kernels.h"

namespace {

struct GldsArgs {
    GemmProblem p[3];
    int M, N, K, lda, ldr, ldc;
    int conv, T_in, T_out, stride, pad, Cin;
    int tiles_m, tiles_n, xcd_remap;
    float out_scale;
    int r_sp16;
};

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int GBM = 128, GBN = 128, GBK = 32;
constexpr int ROWB = 128;                      // bytes per tile row (32 elements x 4 B)
constexpr int STAGE_BYTES = (GBM + GBN) * ROWB;  // 32 KiB

__global__ __launch_bounds__(256) void gemm_nt_split_glds_kernel(const GldsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const GemmProblem pr = a.p[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int rt, ct;
    {
        const int bid = blockIdx.x;
        if (a.xcd_remap) {
            const int x = bid & 7, j = bid >> 3;
            rt = x + 8 * (j / a.tiles_n);
            ct = j % a.tiles_n;
        } else {
            rt = bid / a.tiles_n;
            ct = bid % a.tiles_n;
        }
    }
    const int m0 = rt * GBM, n0 = ct * GBN;

    // ---- DMA coordinates: this wave owns row groups rg = wave*4 + i (8 rows each) of both operand tiles;
    //      lane -> (row = rg*8 + lane/8, physical 16-byte chunk = lane%8)
    const int lrow = lane >> 3, chunk = lane & 7;
    const char* a_src[4];
    const char* w_src[4];
    int a_t0[4];
    bool a_ok[4], w_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + lrow;  // tile-local row
        const int blk = (chunk >> 1) ^ ((r >> 1) & 3);  // logical 32-byte block that must land in this physical slot
        const int col_bytes = blk * 32 + (chunk & 1) * 16;
        const int m = m0 + r;
        a_ok[i] = m < a.M;
        if (a.conv == 1) {
            const int rr = m / a.T_out, to = m - rr * a.T_out;
            a_t0[i] = to * a.stride - a.pad;
            a_src[i] = reinterpret_cast<const char*>(pr.A + (long long)rr * a.T_in * a.Cin) + col_bytes;
        } else {
            a_t0[i] = 0;
            a_src[i] = reinterpret_cast<const char*>(pr.A + (long long)m * a.lda) + col_bytes;
        }
        const int n = n0 + r;
        w_ok[i] = n < a.N;
        w_src[i] = reinterpret_cast<const char*>(pr.W + (long long)n * a.K) + col_bytes;
    }
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    auto issue = [&](int kt, int stage) {
        const int k0 = kt * GBK;
        char* sbase = lds + stage * STAGE_BYTES;
        int kk = 0, ci0 = k0;
        if (a.conv == 1) {
            kk = k0 / a.Cin;          // uniform over the k-tile (Cin % 32 == 0)
            ci0 = k0 - kk * a.Cin;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rg = wave * 4 + i;
            const char* src;
            if (a.conv == 1) {
                const int ti = a_t0[i] + kk;
                src = (a_ok[i] && ti >= 0 && ti < a.T_in) ? a_src[i] + ((long long)ti * a.Cin + ci0) * 4 : zero;
            } else {
                src = a_ok[i] ? a_src[i] + (long long)k0 * 4 : zero;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + rg * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rg = wave * 4 + i;
            const char* src = w_ok[i] ? w_src[i] + (long long)k0 * 4 : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + GBM * ROWB + rg * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment rows of this lane and their swizzle keys
    const int fr = lane & 31, fh = lane >> 5;
    int a_row_off[2], w_row_off[2], a_key[2], w_key[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wr * 64 + i * 32 + fr, rw = wc * 64 + i * 32 + fr;
        a_row_off[i] = ra * ROWB; a_key[i] = (ra >> 1) & 3;
        w_row_off[i] = GBM * ROWB + rw * ROWB; w_key[i] = (rw >> 1) & 3;
    }

    const int nk = a.K / GBK;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, stage ^ 1);
        const char* sbase = lds + stage * STAGE_BYTES;
#pragma unroll
        for (int s16 = 0; s16 < 2; ++s16) {
            const int blk = s16 * 2 + fh;
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* p = sbase + a_row_off[i] + ((blk ^ a_key[i]) << 5);
                ah[i] = *reinterpret_cast<const half8*>(p);
                al[i] = *reinterpret_cast<const half8*>(p + 16);
                const char* q = sbase + w_row_off[i] + ((blk ^ w_key[i]) << 5);
                bh[i] = *reinterpret_cast<const half8*>(q);
                bl[i] = *reinterpret_cast<const half8*>(q + 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile kt+1 has landed (it had the whole MFMA phase)
        __syncthreads();
    }

    // ---- epilogue: the accumulators (one column per lane, 16 scattered rows) go through this wave's 16 KiB of the
    //      now idle stage buffers and leave as whole 16-byte row pieces (16 lanes cover one 256-byte row segment),
    //      instead of 64 strided dword stores per lane.  Bias, output scale and the residual are applied on the way out.
    float* tile = reinterpret_cast<float*>(lds) + wave * (64 * 64);  // [64 rows][64 cols] f32 (256-B rows: b32 writes and b128 reads are conflict-free)
    const int col_l = lane & 31, row_l = (lane >> 5) << 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
                const int col = j * 32 + col_l;
                tile[row * 64 + col] = acc[i][j][r];
            }
    __syncthreads();
    const int c4 = lane & 15;   // 16-byte column piece
    const int rsub = lane >> 4;  // 4 rows per pass
    const int n = n0 + wc * 64 + c4 * 4;
    const bool vec_ok = (a.ldc & 3) == 0 && n + 3 < a.N && (!pr.R || a.r_sp16 || (a.ldr & 3) == 0);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pr.bias) {
        bv.x = n < a.N ? pr.bias[n] : 0.f;
        bv.y = n + 1 < a.N ? pr.bias[n + 1] : 0.f;
        bv.z = n + 2 < a.N ? pr.bias[n + 2] : 0.f;
        bv.w = n + 3 < a.N ? pr.bias[n + 3] : 0.f;
    }
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
        const int row = pass * 4 + rsub;
        const int m = m0 + wr * 64 + row;
        const float4 t = *reinterpret_cast<const float4*>(&tile[row * 64 + c4 * 4]);
        if (m >= a.M) continue;
        float v[4] = {t.x * a.out_scale + bv.x, t.y * a.out_scale + bv.y, t.z * a.out_scale + bv.z, t.w * a.out_scale + bv.w};
        if (pr.R) {
            if (a.r_sp16) {
                // 4 consecutive columns sit in one 8-wide block: hi[4] and lo[4] are two aligned 8-byte loads
                const _Float16* rb = reinterpret_cast<const _Float16*>(pr.R + (long long)m * a.ldr + (n & ~7)) + (n & 4);
                if (n + 3 < a.N) {
                    const half4 h = *reinterpret_cast<const half4*>(rb), l = *reinterpret_cast<const half4*>(rb + 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)h[e] + (float)l[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < a.N) v[e] += (float)rb[e] + (float)rb[8 + e];
                }
            } else if (vec_ok) {
                const float4 rv = *reinterpret_cast<const float4*>(pr.R + (long long)m * a.ldr + n);
                v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < a.N) v[e] += pr.R[(long long)m * a.ldr + n + e];
            }
        }
        if (vec_ok) {
            *reinterpret_cast<float4*>(pr.C + (long long)m * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < a.N) pr.C[(long long)m * a.ldc + n + e] = v[e];
        }
    }
}

}  // namespace

bool gemm_split_glds_supported(const GemmDesc& d) {
    if (d.arith != 1 || d.K % GBK != 0 || d.conv == 2) return false;
    if (d.conv == 1) return d.Cin % GBK == 0;
    return d.lda % 8 == 0;
}

int launch_gemm_split_glds(const GemmDesc& d, hipStream_t s) {
    GldsArgs a;
    for (int i = 0; i < 3; ++i) a.p[i] = d.p[i < d.nprob ? i : 0];
    a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldr = d.ldr; a.ldc = d.ldc;
    a.conv = d.conv; a.T_in = d.T_in; a.T_out = d.T_out; a.stride = d.stride; a.pad = d.pad; a.Cin = d.Cin;
    a.tiles_m = (d.M + GBM - 1) / GBM;
    a.tiles_n = (d.N + GBN - 1) / GBN;
    a.xcd_remap = (a.tiles_m % 8 == 0) ? 1 : 0;
    a.out_scale = d.out_scale != 0.f ? d.out_scale : 1.f;
    a.r_sp16 = d.r_sp16;
    constexpr size_t lds = 2 * (size_t)STAGE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_split_glds_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_nt_split_glds_kernel, dim3(a.tiles_m * a.tiles_n, 1, d.nprob), dim3(256), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
