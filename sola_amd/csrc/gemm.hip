// f32 "NT" GEMM on the gfx950 matrix cores: C[M,N] = A[M,K] * W[N,K]^T + bias (+ residual).
//
// Replaces F.linear at tools/attention.py:63-65,73 and, through an implicit im2col gather of the channels-last
// activations, F.conv1d at module/ws.py:14-22 (k=3 window = 3*Cin contiguous floats per output row, zero-filled at
// the sequence edges).  Exact f32 arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; there is no
// xf32/TF32 on gfx950), so results are an fmaf chain per output element.
//
// Tiling (wave64): 256 threads = 4 waves in a 2x2 grid; block tile BM x BN x 32, each wave owns (BM/2) x (BN/2) as
// 32x32 MFMA tiles (16 accumulator registers each).  Operands are staged global -> VGPR -> LDS with 16-byte
// accesses; the LDS row pitch of 36 floats makes the per-lane ds_read_b128 fragment reads conflict-free
// (16-lane groups hit 16 distinct 16-byte slots).  One ds_read_b128 per lane feeds four MFMAs: lanes 0-31 take
// k = 8s+{0..3}, lanes 32-63 take k = 8s+{4..7}; operands A and B use the same k permutation so the sum is unchanged.
// Double-buffered LDS, one barrier per 32-deep k-tile; the next tile's global loads are in flight during the MFMAs.
// Up to three problems that share all dimensions run in one launch (q/k/v projections) via blockIdx.z.
#include <algorithm>

#include "kernels.h"

namespace {

struct GemmArgs {
    GemmProblem p[3];
    int M, N, K, lda, ldr, ldc;
    int conv, T_in, T_out, stride, pad, Cin;
    const int2* rowmap;  // conv 1, ragged batches (GemmDesc::rowmap)
    int tiles_m, tiles_n, xcd_remap;
    float out_scale;  // result multiplier (power of two undoing the weight pre-scale of the split-f16 path)
    const float* out_scale_dev;  // optional further multiplier in device memory (GemmDesc::out_scale_dev)
    int r_sp16;       // residual R is stored as split-f16 pairs
    int c_sp16;       // C is written as split-f16 pairs
    int ksplit;       // > 1: blockIdx.y owns k-tiles [y*kt_per, (y+1)*kt_per) and writes raw partial sums to `part`
    int kt_per;
    float* part;      // [nprob][ksplit][M][N]
    int* guard;       // c_sp16: range guard word (GemmDesc::guard)
    const float* w_nn[3];  // few-row shape, NN form (GemmDesc::w_nn)
    int w_nn_rows;
};

// |v| must stay inside the f16 range to be written as a split-f16 pair; NaN fails the comparison too
__device__ __forceinline__ void guard_sp16(int* guard, float m) {
    if (guard && !(m < 65000.f)) atomicOr(guard, 1);
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;
constexpr int LDP = 36;  // LDS row pitch in floats (144 B: 16-byte aligned, conflict-free for b128 fragment reads)

// ARITH 0: operands are f32, v_mfma_f32_32x32x2_f32 (exact).
// ARITH 1: operands are "split-f16" rows - every 8 consecutive f32 values x are stored in the same 32 bytes as
//          [8 x f16 hi | 8 x f16 lo] with hi = f16(x), lo = f16(x - hi) - and each product runs as three
//          v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi, f32 accumulate): ~22-bit products at 3/16 of the f32 MFMA
//          cost.  Bytes per element, tile staging and LDS layout are identical to ARITH 0 (a k16 MFMA step consumes
//          two 32-byte blocks; lanes 0-31 take the first, lanes 32-63 the second - exactly the MFMA A/B fragment).
// NW = 8 (round 4, exact f32 at 128x128): eight waves of 32x64 instead of four of 64x64 - half the accumulators per wave (the two-level sum
// doubles them), <= 128 registers, so the CU's two blocks put FOUR waves on every SIMD instead of two.  PMC on the four-wave form
// (profiles/r04_gemm_f32_pmc.txt): matrix pipe busy 70.6 %, not power-bound - with two waves per SIMD, each from another block, both are
// regularly at their k-tile barrier at once.  Same k order per output element: bit-identical results.
template <int BM, int BN, int PIPE, int ARITH, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW / 2)  // (hipcc: the second argument is waves per SIMD)
void gemm_nt_f32_kernel(const GemmArgs a) {  // two blocks per CU (their LDS allows exactly that): NW / 2 waves per SIMD, <= 1024 / NW registers
    constexpr int LROWS = 8 * NW;                       // rows one load pass of the block covers (8 lanes per 128-byte row segment)
    constexpr int RA = BM / LROWS, RW = BN / LROWS;     // 16-byte loads per thread per operand per k-tile
    constexpr int WROWS = NW / 2;                       // wave grid: WROWS x 2
    constexpr int TM = BM / (32 * WROWS), TN = BN / 64; // 32x32 MFMA tiles per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BM][LDP]
    float* Ws = smem + 2 * BM * LDP;   // [2][BN][LDP]

    const GemmProblem pr = a.p[blockIdx.z];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // block -> tile.  With 8 XCDs (block b runs on XCD b % 8) keep all column tiles of one row panel on one XCD so
    // the A panel is fetched into a single L2.
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
    const int m0 = rt * BM, n0 = ct * BN;

    // ---- per-thread load coordinates: 8 lanes cover one 32-float (128 B) row segment
    const int lr = tid >> 3;
    const int lk = (tid & 7) << 2;
    long long a_off[RA];
    int a_t0[RA];
    bool a_ok[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + lr + LROWS * i;
        a_ok[i] = m < a.M;
        if (a.conv == 1 && a.rowmap) {
            // ragged: a_t0 = tap-validity bits, a_off = offset of the window's tap 0 (possibly outside the sequence)
            const int2 rm = a.rowmap[a_ok[i] ? m : a.M - 1];
            a_t0[i] = rm.y;
            a_off[i] = (long long)rm.x * a.Cin;
        } else if (a.conv == 1) {
            const int r = m / a.T_out;
            const int to = m - r * a.T_out;
            a_t0[i] = to * a.stride - a.pad;
            a_off[i] = (long long)r * a.T_in * a.Cin;
        } else if (a.conv == 2) {
            // transposed conv (dX): output row = (r, ti) of the conv INPUT; source rows are conv OUTPUT rows
            // (r, to) with to * stride - pad + kk == ti.  Here T_out = rows per r of the result (= conv T_in),
            // T_in = rows per r of the source (= conv T_out), Cin = source row width (= conv Cout).
            const int r = m / a.T_out;
            const int ti = m - r * a.T_out;
            a_t0[i] = ti + a.pad;
            a_off[i] = (long long)r * a.T_in * a.Cin;
        } else {
            a_t0[i] = 0;
            a_off[i] = (long long)m * a.lda;
        }
    }
    long long w_off[RW];
    bool w_ok[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int n = n0 + lr + LROWS * j;
        w_ok[j] = n < a.N;
        w_off[j] = (long long)n * a.K;
    }

    float4 ra[RA], rw[RW];
    auto load_tile = [&](int k0) {
        const int k = k0 + lk;
        const bool kok = k < a.K;
        int kk = 0, ci = k;
        if (a.conv) {
            kk = k / a.Cin;
            ci = k - kk * a.Cin;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            bool ok = a_ok[i] && kok;
            const float* src;
            if (a.conv == 1 && a.rowmap) {
                ok = ok && ((a_t0[i] >> kk) & 1);
                src = pr.A + a_off[i] + (long long)kk * a.Cin + ci;
            } else if (a.conv == 1) {
                const int ti = a_t0[i] + kk;
                ok = ok && ti >= 0 && ti < a.T_in;
                src = pr.A + a_off[i] + (long long)ti * a.Cin + ci;
            } else if (a.conv == 2) {
                const int num = a_t0[i] - kk;  // = to * stride
                const int to = num / a.stride;
                ok = ok && num >= 0 && to * a.stride == num && to < a.T_in;
                src = pr.A + a_off[i] + (long long)to * a.Cin + ci;
            } else {
                src = pr.A + a_off[i] + k;
            }
            ra[i] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const bool ok = w_ok[j] && kok;
            rw[j] = ok ? *reinterpret_cast<const float4*>(pr.W + w_off[j] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i)
            *reinterpret_cast<float4*>(&As[(buf * BM + lr + LROWS * i) * LDP + lk]) = ra[i];
#pragma unroll
        for (int j = 0; j < RW; ++j)
            *reinterpret_cast<float4*>(&Ws[(buf * BN + lr + LROWS * j) * LDP + lk]) = rw[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Exact-f32 arithmetic only: the MFMA chain rounds after every product, so one accumulator summed over K = 1024 ... 3072
    // carries a random walk of ~sqrt(K/2) roundings of the running sum (the 256-sample batch sat 9.9e-4 from the fp32 oracle, the
    // split-f16 mode - whose f16 MFMA rounds once per 16 products - 4.3e-4).  Two levels instead: `acc` collects FOLD k-tiles (128
    // values of k), then is added to `outer` and cleared - sqrt(64) + sqrt(K/128) roundings instead of sqrt(K/2), in a fixed order
    // (deterministic), for 64 extra VGPRs (the kernel runs two waves per SIMD on its LDS footprint: 256 are available) and one
    // v_add per 32 MFMAs.
    constexpr int FOLD = 4;
    f32x16 outer[ARITH == 0 ? TM : 1][ARITH == 0 ? TN : 1];
    if constexpr (ARITH == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) outer[i][j][r] = 0.f;
    }
    auto fold = [&](int done_tiles, bool last) {  // after `done_tiles` k-tiles of this block's range
        if constexpr (ARITH == 0) {
            if (last || done_tiles % FOLD == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            outer[i][j][r] += acc[i][j][r];
                            acc[i][j][r] = last ? outer[i][j][r] : 0.f;
                        }
            }
        }
    };

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) << 2;
    const int nk = (a.K + BK - 1) / BK;

    auto compute = [&](int buf, int ks_begin, int ks_end) {
        if constexpr (ARITH == 1) {
            const char* Ab = reinterpret_cast<const char*>(&As[(buf * BM + wr * (TM * 32) + frag_row) * LDP]) + (lane >> 5) * 32;
            const char* Wb = reinterpret_cast<const char*>(&Ws[(buf * BN + wc * (BN / 2) + frag_row) * LDP]) + (lane >> 5) * 32;
#pragma unroll
            for (int s16 = ks_begin / 2; s16 < ks_end / 2; ++s16) {
                half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const char* p = Ab + i * 32 * LDP * 4 + s16 * 64;
                    ah[i] = *reinterpret_cast<const half8*>(p);
                    al[i] = *reinterpret_cast<const half8*>(p + 16);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const char* p = Wb + j * 32 * LDP * 4 + s16 * 64;
                    bh[j] = *reinterpret_cast<const half8*>(p);
                    bl[j] = *reinterpret_cast<const half8*>(p + 16);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
            return;
        }
        const float* Ab = &As[(buf * BM + wr * (TM * 32) + frag_row) * LDP + frag_k];
        const float* Wb = &Ws[(buf * BN + wc * (BN / 2) + frag_row) * LDP + frag_k];
        // the fragments of step ks + 1 are requested before the MFMAs of step ks (pinned by a scheduling barrier: left alone the scheduler
        // issues reads -> wait -> MFMAs per step, with the LDS latency in front of every step's first MFMA)
        float4 af[2][TM], bf[2][TN];
        auto frag = [&](int ks, int slot) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[slot][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDP + ks * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const float4*>(Wb + j * 32 * LDP + ks * 8);
        };
        frag(ks_begin, 0);
#pragma unroll
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            const int slot = (ks - ks_begin) & 1;
            if (ks + 1 < ks_end) frag(ks + 1, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = s == 0 ? af[slot][i].x : s == 1 ? af[slot][i].y : s == 2 ? af[slot][i].z : af[slot][i].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float bv = s == 0 ? bf[slot][j].x : s == 1 ? bf[slot][j].y : s == 2 ? bf[slot][j].z : bf[slot][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };

    // split-K: this block's share of the k-tiles
    const int ktb = a.ksplit > 1 ? (int)blockIdx.y * a.kt_per : 0;
    const int kte = a.ksplit > 1 ? min(nk, ktb + a.kt_per) : nk;
    if constexpr (PIPE == 0) {
        // simple schedule: loads of tile k+1 in flight during tile k; LDS store + barrier at the tile boundary
        load_tile(ktb * BK);
        store_tile(0);
        __syncthreads();
        for (int kt = ktb; kt < kte; ++kt) {
            const int buf = (kt - ktb) & 1;
            if (kt + 1 < kte) load_tile((kt + 1) * BK);
            compute(buf, 0, BK / 8);
            fold(kt - ktb + 1, kt + 1 == kte);
            if (kt + 1 < kte) store_tile(buf ^ 1);
            __syncthreads();
        }
    } else {
        // deeper schedule: the registers always hold tile k+1 on entry; its LDS store and the global loads of tile
        // k+2 are issued in the MIDDLE of tile k's MFMA stream, so the tile boundary is a bare barrier + fragment read
        load_tile(ktb * BK);
        store_tile(0);
        if (ktb + 1 < kte) load_tile((ktb + 1) * BK);
        __syncthreads();
        for (int kt = ktb; kt < kte; ++kt) {
            const int buf = (kt - ktb) & 1;
            compute(buf, 0, BK / 16);
            if (kt + 1 < kte) store_tile(buf ^ 1);
            if (kt + 2 < kte) load_tile((kt + 2) * BK);
            compute(buf, BK / 16, BK / 8);
            fold(kt - ktb + 1, kt + 1 == kte);
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int col_l = lane & 31;
    const int row_l = (lane >> 5) << 2;
    if (a.ksplit > 1) {  // raw partial sums; splitk_reduce_kernel applies scale, bias, residual and the output format
        float* part = a.part + ((long long)blockIdx.z * a.ksplit + blockIdx.y) * a.M * a.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wc * (BN / 2) + j * 32 + col_l;
            if (n >= a.N) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
                    if (m < a.M) part[(long long)m * a.N + n] = acc[i][j][r];
                }
        }
        return;
    }
    const float osc = (a.out_scale_dev ? a.out_scale * *a.out_scale_dev : a.out_scale) * (pr.scale_dev ? *pr.scale_dev : 1.f);
    if (ARITH == 0 && pr.R && !a.r_sp16 && !a.c_sp16) {
        // f32 residual, f32 output (out-projections, the dX GEMMs that add the skip gradient): the generic loop below compiles to load -> full
        // wait -> add -> store per ELEMENT (round 4 disassembly: 16 TM TN serialized round trips per lane, 13 % of such a launch).  Here the
        // tile's residual values are requested first - clamped coordinates, unconditional loads, back to back - and then added and stored.
        float rv[TM][TN][16];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = min(n0 + wc * (BN / 2) + j * 32 + col_l, a.N - 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = min(m0 + wr * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + row_l, a.M - 1);
                    rv[i][j][r] = pr.R[(long long)m * a.ldr + n];
                }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wc * (BN / 2) + j * 32 + col_l;
            const float bv = (pr.bias && n < a.N) ? pr.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wr * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
                    if (n < a.N && m < a.M) pr.C[(long long)m * a.ldc + n] = (acc[i][j][r] * osc + bv) + rv[i][j][r];  // same association as below
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wc * (BN / 2) + j * 32 + col_l;
        if (n >= a.N) continue;
        const float bv = pr.bias ? pr.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
                if (m < a.M) {
                    float v = acc[i][j][r] * osc + bv;
                    if (pr.R) {
                        if (a.r_sp16) {  // residual kept as split-f16 pairs: element n sits in block n/8 as hi[n%8], lo[n%8]
                            const _Float16* rb = reinterpret_cast<const _Float16*>(pr.R + (long long)m * a.ldr + (n & ~7));
                            v += (float)rb[n & 7] + (float)rb[8 + (n & 7)];
                        } else {
                            v += pr.R[(long long)m * a.ldr + n];
                        }
                    }
                    if (a.c_sp16) {  // element n of block n/8: hi[n%8] | lo[n%8]
                        _Float16* cb = reinterpret_cast<_Float16*>(pr.C + (long long)m * a.ldc + (n & ~7));
                        _Float16 hi, lo;
                        split_f16(v, hi, lo);
                        cb[n & 7] = hi;
                        cb[8 + (n & 7)] = lo;
                        guard_sp16(a.guard, fabsf(v));
                    } else {
                        pr.C[(long long)m * a.ldc + n] = v;
                    }
                }
            }
        }
    }
}

// Second pass of the split-K: C = out_scale * (sum over the S partials, in index order) + bias (+ R), 4 columns per thread.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs a) {
    const GemmProblem pr = a.p[blockIdx.z];
    const int n4 = a.N >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.M * n4) return;
    const int m = (int)(i / n4), n = (int)(i - (long long)m * n4) * 4;
    const float* part = a.part + (long long)blockIdx.z * a.ksplit * a.M * a.N + (long long)m * a.N + n;
    float4 sum = *reinterpret_cast<const float4*>(part);
    for (int sidx = 1; sidx < a.ksplit; ++sidx) {
        const float4 v = *reinterpret_cast<const float4*>(part + (long long)sidx * a.M * a.N);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const float osc = (a.out_scale_dev ? a.out_scale * *a.out_scale_dev : a.out_scale) * (pr.scale_dev ? *pr.scale_dev : 1.f);
    float v[4] = {sum.x * osc, sum.y * osc, sum.z * osc, sum.w * osc};
    if (pr.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += pr.bias[n + e];
    }
    if (pr.R) {
        if (a.r_sp16) {
            const _Float16* rb = reinterpret_cast<const _Float16*>(pr.R + (long long)m * a.ldr + (n & ~7)) + (n & 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rb[e] + (float)rb[8 + e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += pr.R[(long long)m * a.ldr + n + e];
        }
    }
    if (a.c_sp16) {
        _Float16* cb = reinterpret_cast<_Float16*>(pr.C + (long long)m * a.ldc + (n & ~7)) + (n & 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 hi, lo;
            split_f16(v[e], hi, lo);
            cb[e] = hi;
            cb[8 + e] = lo;
        }
        guard_sp16(a.guard, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) pr.C[(long long)m * a.ldc + n + e] = v[e];
    }
}

// ---- few rows (round 4): one sample per call or per optimizer step has 256-2000 token rows -----------------------------------
// The 64x64 shape above needs a split-K over 8 ranges to fill the chip at M = 256 (512 blocks of four k-tiles each: load -> LDS ->
// barrier -> MFMA with nothing to overlap, 15 us) and a second launch to fold the ranges (5 us): 41 such pairs are 0.8 ms of a 2.1 ms
// one-sample training step.  Here a block owns a 32 x 32 output tile (M = 256, N = 1024: 256 tiles, one per CU) over the WHOLE of K,
// and the split over K happens inside the block: wave w reduces K range w of four through a double-buffered stage in a slice of LDS
// that only it touches (no block barrier in the loop: LDS executes one wave's writes and reads in order), the four 32 x 32 partial
// sums meet in LDS once, are added in wave order (fixed: deterministic) and leave through the usual epilogue (scale, bias,
// residual) as 16-byte row stores.  No partial sums in memory, no second launch.  Logical tiles are ordered W-panel-major and dealt
// to the XCDs in contiguous ranges, so one XCD's L2 holds N/8 rows of W and all of A.
// Exact f32 (v_mfma_f32_32x32x2_f32), plain rows only (no conv gather), N % 32 == 0, K % 128 == 0.
constexpr int SM_LD = 36;  // stage row pitch in floats (as LDP)
// NW waves split K: 4 (two blocks per CU) or 8 (launches of at most ~one tile per CU: twice the loads in flight per tile, 13 -> ~9 us
// at 256 x 1024 x 1024 where a wave's four-to-eight-step loop is bound by the latency of its own loads)
// CONV: A is the channels-last conv input and K = k * Cin (implicit im2col, GemmArgs::conv == 1; Cin % 32 == 0, so a 32-deep k-step lies
// inside one tap): a row's window is (first tap's address, valid-tap bits) - from the uniform geometry or the ragged row map - and a step
// reads its tap's 32 channels, zeros where the tap falls outside the sequence (clamped address, value select).
// NN (round 5): W is given as its row-major [K][N] image (GemmDesc::w_nn: up to three stacked matrices) - a step's W tile is then 32
// reduction rows x 32 output columns (rows of 128 contiguous bytes, the same load shape), staged as it lies, and the B fragments are column
// reads (16 ds_read_b32 per step instead of 4 ds_read_b128; conflict-free: 32 consecutive columns per half-wave).  The same products in the
// same order as the NT form on a transposed copy: bit-identical, without the copy (the 49 weight transpositions of a one-sample step).
template <int NW, bool CONV, bool NN = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void gemm_nt_f32_small_kernel(const GemmArgs a) {
    static_assert(!(CONV && NN), "NN: plain rows only");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    GemmProblem pr = a.p[0];  // (a.p[blockIdx.z] made the compiler copy the argument array to scratch to index it)
    if (blockIdx.z == 1) pr = a.p[1];
    if (blockIdx.z == 2) pr = a.p[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = a.tiles_m * a.tiles_n;
    const int bid = blockIdx.x;
    const int L = a.xcd_remap ? (bid & 7) * (total >> 3) + (bid >> 3) : bid;  // W-panel-major logical tile
    const int ct = L / a.tiles_m, rt = L - ct * a.tiles_m;
    const int m0 = rt * 32, n0 = ct * 32;
    float* const st = smem + wave * (2 * 2 * 32 * SM_LD);  // this wave's two stages of (A 32 x 32 | W 32 x 32)
    const int kw = a.K / NW;                               // this wave's K range
    const int kb = wave * kw;
    // loads: 8 lanes cover one 128-byte row segment, 8 rows per instruction, 4 instructions per operand and step
    const int lr = lane >> 3, lk = (lane & 7) << 2;
    const float* ap[4];
    const float* wp[4];
    int amask[4], afall[4];  // CONV: valid taps of the row's window, and one of them (the address of an invalid tap's load)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = min(m0 + lr + 8 * i, a.M - 1);  // rows past M: clamped (their results are never stored)
        if constexpr (CONV) {
            if (a.rowmap) {
                const int2 rm = a.rowmap[m];
                ap[i] = pr.A + (long long)rm.x * a.Cin + lk;
                amask[i] = rm.y;
            } else {
                const int r = m / a.T_out, to = m - r * a.T_out, t0 = to * a.stride - a.pad;
                ap[i] = pr.A + ((long long)r * a.T_in + t0) * a.Cin + lk;
                int bits = 0;
                for (int kk = 0; kk < a.K / a.Cin; ++kk) bits |= (t0 + kk >= 0 && t0 + kk < a.T_in) ? 1 << kk : 0;
                amask[i] = bits;
            }
            afall[i] = amask[i] ? __builtin_ctz(amask[i]) : 0;
        } else {
            ap[i] = pr.A + (long long)m * a.lda + kb + lk;
            amask[i] = afall[i] = 0;
        }
        wp[i] = NN ? nullptr : pr.W + (long long)(n0 + lr + 8 * i) * a.K + kb + lk;
    }
    // NN: element (reduction row kb + k + lr + 8 i, column n0 + lk) of the stacked matrices; a 32-row step lies inside one of them
    auto nn_row = [&](int k) -> const float* {
        const int kabs = kb + k, seg = kabs / a.w_nn_rows;  // wave-uniform
        const float* base = seg == 0 ? a.w_nn[0] : (seg == 1 ? a.w_nn[1] : a.w_nn[2]);
        return base + (long long)(kabs - seg * a.w_nn_rows + lr) * a.N + n0 + lk;
    };
    // (prefetch registers as eight named values and the staging as macros: with arrays captured by lambdas the compiler kept them in a
    // 144-byte scratch frame - a scratch store behind every load and a full wait in front of every LDS write)
    float4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define SM_LOAD_A(i, k, dst)                                                                                             \
    do {                                                                                                                \
        if constexpr (CONV) {                                                                                           \
            const int kabs_ = kb + (k), kk_ = kabs_ / a.Cin, ci_ = kabs_ - kk_ * a.Cin; /* wave-uniform */              \
            const bool ok_ = (amask[i] >> kk_) & 1;                                                                     \
            const float4 v_ = *reinterpret_cast<const float4*>(ap[i] + (long long)(ok_ ? kk_ : afall[i]) * a.Cin + ci_); \
            dst = ok_ ? v_ : z4;                                                                                        \
        } else {                                                                                                        \
            dst = *reinterpret_cast<const float4*>(ap[i] + (k));                                                         \
        }                                                                                                               \
    } while (0)
#define SM_LOAD_W(i, k, dst, nnp)                                                                                        \
    do {                                                                                                                \
        if constexpr (NN) dst = *reinterpret_cast<const float4*>((nnp) + (long long)(8 * (i)) * a.N);                    \
        else dst = *reinterpret_cast<const float4*>(wp[i] + (k));                                                        \
    } while (0)
#define SM_LOAD(k)                                                                                                      \
    do {                                                                                                                \
        const float* nnp_ = nullptr;                                                                                    \
        if constexpr (NN) nnp_ = nn_row(k);                                                                             \
        SM_LOAD_A(0, k, ra0); SM_LOAD_W(0, k, rw0, nnp_);                                                                \
        SM_LOAD_A(1, k, ra1); SM_LOAD_W(1, k, rw1, nnp_);                                                                \
        SM_LOAD_A(2, k, ra2); SM_LOAD_W(2, k, rw2, nnp_);                                                                \
        SM_LOAD_A(3, k, ra3); SM_LOAD_W(3, k, rw3, nnp_);                                                                \
    } while (0)
#define SM_STORE(buf)                                                                                                   \
    do {                                                                                                                \
        float* As_ = st + (buf) * (2 * 32 * SM_LD) + lr * SM_LD + lk;                                                    \
        float* Ws_ = As_ + 32 * SM_LD;                                                                                   \
        *reinterpret_cast<float4*>(As_) = ra0; *reinterpret_cast<float4*>(As_ + 8 * SM_LD) = ra1;                         \
        *reinterpret_cast<float4*>(As_ + 16 * SM_LD) = ra2; *reinterpret_cast<float4*>(As_ + 24 * SM_LD) = ra3;            \
        *reinterpret_cast<float4*>(Ws_) = rw0; *reinterpret_cast<float4*>(Ws_ + 8 * SM_LD) = rw1;                         \
        *reinterpret_cast<float4*>(Ws_ + 16 * SM_LD) = rw2; *reinterpret_cast<float4*>(Ws_ + 24 * SM_LD) = rw3;            \
    } while (0)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int frag_row = lane & 31, frag_k = (lane >> 5) << 2;
    const int nk = kw >> 5;
    SM_LOAD(0);
    SM_STORE(0);
    if (nk > 1) SM_LOAD(32);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const float* Ab = st + buf * (2 * 32 * SM_LD) + frag_row * SM_LD + frag_k;
        const float* Wb = Ab + 32 * SM_LD;
        const float4 af0 = *reinterpret_cast<const float4*>(Ab), af1 = *reinterpret_cast<const float4*>(Ab + 8);
        const float4 af2 = *reinterpret_cast<const float4*>(Ab + 16), af3 = *reinterpret_cast<const float4*>(Ab + 24);
        float4 bf0, bf1, bf2, bf3;
        if constexpr (NN) {  // the staged tile is [reduction row][column]: B[k = frag_k + 8 j + e][n = frag_row]
            const float* Wc = st + buf * (2 * 32 * SM_LD) + 32 * SM_LD + frag_k * SM_LD + frag_row;
            bf0 = make_float4(Wc[0], Wc[SM_LD], Wc[2 * SM_LD], Wc[3 * SM_LD]);
            bf1 = make_float4(Wc[8 * SM_LD], Wc[9 * SM_LD], Wc[10 * SM_LD], Wc[11 * SM_LD]);
            bf2 = make_float4(Wc[16 * SM_LD], Wc[17 * SM_LD], Wc[18 * SM_LD], Wc[19 * SM_LD]);
            bf3 = make_float4(Wc[24 * SM_LD], Wc[25 * SM_LD], Wc[26 * SM_LD], Wc[27 * SM_LD]);
        } else {
            bf0 = *reinterpret_cast<const float4*>(Wb); bf1 = *reinterpret_cast<const float4*>(Wb + 8);
            bf2 = *reinterpret_cast<const float4*>(Wb + 16); bf3 = *reinterpret_cast<const float4*>(Wb + 24);
        }
        if (kt + 1 < nk) SM_STORE(buf ^ 1);          // the registers hold step kt + 1
        // the stage written here is read by other lanes of THIS wave in the next iteration: pin the order (a wave's DS operations execute in
        // order; this keeps the compiler from ever moving the next iteration's fragment reads above these writes - no instruction is emitted)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (kt + 2 < nk) SM_LOAD((kt + 2) * 32);
#define SM_MFMA4(af, bf)                                                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);         \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        SM_MFMA4(af0, bf0) SM_MFMA4(af1, bf1) SM_MFMA4(af2, bf2) SM_MFMA4(af3, bf3)
    }
#undef SM_LOAD
#undef SM_LOAD_W
#undef SM_LOAD_A
#undef SM_STORE
#undef SM_MFMA4
    // the four K ranges meet: C/D layout of the 32x32 MFMA - col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    __syncthreads();  // every wave is done with its stages
    float* red = smem;  // [NW][32][33]
    {
        const int col = lane & 31, row_l = (lane >> 5) << 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + row_l) * 33 + col] = acc[r];
    }
    __syncthreads();
    if (tid >= 256) return;  // NW = 8: the first four waves write the tile
    const int row = tid >> 3, c4 = (tid & 7) << 2;
    const int m = m0 + row, n = n0 + c4;
    if (m >= a.M) return;
    const float osc = (a.out_scale_dev ? a.out_scale * *a.out_scale_dev : a.out_scale) * (pr.scale_dev ? *pr.scale_dev : 1.f);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float* q = red + row * 33 + c4 + e;
        float t = q[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) t += q[w * 32 * 33];  // wave order: fixed
        v[e] = t * osc;
    }
    if (pr.bias) {
        const float4 b = *reinterpret_cast<const float4*>(pr.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (pr.R) {
        const float4 r4 = *reinterpret_cast<const float4*>(pr.R + (long long)m * a.ldr + n);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
    }
    *reinterpret_cast<float4*>(pr.C + (long long)m * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
}

int g_gemm_small_rows = 2048;  // sola_tune "gemm_small_rows": exact-f32 GEMMs of at most this many rows take the 32x32 in-block split-K shape (0 = never)
static bool gemm_small_applies(const GemmDesc& d) {
    if (d.r_sp16 || d.c_sp16 || d.gn_gamma) return false;  // the kernel reads R and writes C as plain f32 rows, no fused norm
    if (d.w_nn_rows) {  // NN form: one problem, plain rows, whole 32-row steps inside each stacked matrix, 16-byte aligned rows
        if (d.nprob != 1 || d.conv || d.w_nn_rows % 32 != 0 || d.K % d.w_nn_rows != 0 || d.K / d.w_nn_rows > 3 || d.N % 4 != 0) return false;
        for (int j = 0; j < d.K / d.w_nn_rows; ++j)
            if (!d.w_nn[j] || (reinterpret_cast<uintptr_t>(d.w_nn[j]) & 15)) return false;
    }
    if (d.arith != 0 || d.conv > 1 || d.M > g_gemm_small_rows || d.N % 32 != 0 || d.K % 128 != 0 || d.ldc % 4 != 0) return false;
    if (d.conv ? (d.Cin % 32 != 0 || d.K % d.Cin != 0 || d.K / d.Cin > 8 || (!d.rowmap && (d.T_out <= 0 || d.T_in <= 0))) : d.lda % 4 != 0) return false;
    for (int j = 0; j < d.nprob; ++j) {
        if (d.p[j].R && d.ldr % 4 != 0) return false;
        if (d.p[j].bias && (reinterpret_cast<uintptr_t>(d.p[j].bias) & 15)) return false;
        if ((reinterpret_cast<uintptr_t>(d.p[j].A) | (d.w_nn_rows ? 0 : reinterpret_cast<uintptr_t>(d.p[j].W)) | reinterpret_cast<uintptr_t>(d.p[j].C)) & 15) return false;
        if (d.p[j].R && (reinterpret_cast<uintptr_t>(d.p[j].R) & 15)) return false;
    }
    return true;
}
int g_gemm_small_nw8 = 1;  // sola_tune "gemm_small_nw8": 0 = always four waves per tile (A/B)
template <int NW, bool CONV, bool NN = false>
static int launch_small_n(const GemmArgs& a, int nprob, hipStream_t s) {
    constexpr size_t lds = (size_t)NW * 2 * 2 * 32 * SM_LD * sizeof(float);  // NW waves x two stages (73.7 / 147.5 KB); the reduction reuses it
    static_assert(NW * 32 * 33 <= NW * 2 * 2 * 32 * SM_LD, "the reduction tile must fit in the stages");
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_small_kernel<NW, CONV, NN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.done(dev);
    }
    hipLaunchKernelGGL((gemm_nt_f32_small_kernel<NW, CONV, NN>), dim3(a.tiles_m * a.tiles_n, 1, nprob), dim3(64 * NW), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
static int launch_small(const GemmArgs& base, int nprob, hipStream_t s) {
    GemmArgs a = base;
    a.tiles_m = (a.M + 31) / 32;
    a.tiles_n = a.N / 32;
    a.xcd_remap = ((a.tiles_m * a.tiles_n) % 8 == 0) ? 1 : 0;
    const long long tiles = (long long)a.tiles_m * a.tiles_n * nprob;
    const bool nw8 = g_gemm_small_nw8 && a.K % 256 == 0 && tiles <= 2 * sola_cu_count();
    if (a.w_nn_rows) return nw8 ? launch_small_n<8, false, true>(a, nprob, s) : launch_small_n<4, false, true>(a, nprob, s);
    if (a.conv) return nw8 ? launch_small_n<8, true>(a, nprob, s) : launch_small_n<4, true>(a, nprob, s);
    return nw8 ? launch_small_n<8, false>(a, nprob, s) : launch_small_n<4, false>(a, nprob, s);
}

template <int BM, int BN, int PIPE, int ARITH, int NW = 4>
int launch_tile(const GemmArgs& base, int nprob, hipStream_t s) {
    GemmArgs a = base;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    a.xcd_remap = (a.tiles_m % 8 == 0) ? 1 : 0;
    constexpr size_t lds = (size_t)(BM + BN) * 2 * LDP * sizeof(float);
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_kernel<BM, BN, PIPE, ARITH, NW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.done(dev);
    }
    dim3 grid(a.tiles_m * a.tiles_n, a.ksplit > 1 ? a.ksplit : 1, nprob);
    hipLaunchKernelGGL((gemm_nt_f32_kernel<BM, BN, PIPE, ARITH, NW>), grid, dim3(64 * NW), lds, s, a);
    SOLA_LAUNCH_CHECK();
    if (a.ksplit > 1) {
        const long long quads = (long long)a.M * (a.N >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256), 1, nprob), dim3(256), 0, s, a);
        SOLA_LAUNCH_CHECK();
    }
    return SOLA_OK;
}

int g_gemm_splitk = 1;  // 0 disables the split-K of small grids (A/B)
int g_gemm_f32_nw8 = 1;         // sola_tune "gemm_f32_nw8": the exact-f32 128x128 shape with eight waves per block (0 = four; A/B)
int g_gemm_splitk_max = 8;      // sola_tune "gemm_splitk_max": most K ranges per tile (A/B)
int g_gemm_splitk_tiles = 512;  // grids with fewer 64x64 tiles than this are split (target: twice as many blocks)
int g_gemm_variant = -1;  // -1 auto (measured: simple schedule wins on 128x128 by 5%, mid-tile staging on 64x64 by 6%), 0 / 1 force

}  // namespace

bool gemm_nn_supported(const GemmDesc& d) { return d.w_nn_rows > 0 && gemm_small_applies(d); }

int g_gemm_glds = 3;  // split-f16 GEMM, direct-to-LDS staging (gemm_glds.hip): 0 off, 1 128x128 blocks, 4 256x256 blocks, 3 auto
int g_gemm_glds_force = 0;  // tests: take the direct-to-LDS kernels for grids of any size
void sola_gemm_set_glds_force(int v) { g_gemm_glds_force = v; }
void sola_gemm_set_variant(int v) { g_gemm_variant = v; }
void sola_gemm_set_splitk(int v) { g_gemm_splitk = v != 0; if (v > 1) g_gemm_splitk_tiles = v; }
void sola_gemm_set_splitk_max(int v) { g_gemm_splitk_max = v < 2 ? 2 : v; }
void sola_gemm_set_small_rows(int v) { g_gemm_small_rows = v; }
void sola_gemm_set_small_nw8(int v) { g_gemm_small_nw8 = v; }
void sola_gemm_set_f32_nw8(int v) { g_gemm_f32_nw8 = v; }
void sola_gemm_set_glds(int v) { g_gemm_glds = v; }
bool gemm_split_glds_supported(const GemmDesc& d);
int launch_splitk_reduce(const float* part, int ksplit, int nprob, float* const* C, int M, int N, int ldc, const float* out_scale_dev,
                         const float* scale_dev, hipStream_t s) {
    SOLA_ARG(part && ksplit >= 1 && nprob >= 1 && nprob <= 3 && N % 4 == 0 && ldc % 4 == 0, "splitk_reduce: ksplit %d nprob %d N %d", ksplit, nprob, N);
    GemmArgs a{};
    for (int j = 0; j < 3; ++j) {
        a.p[j] = GemmProblem{};
        a.p[j].C = C[j < nprob ? j : 0];
        a.p[j].scale_dev = scale_dev;
    }
    a.M = M; a.N = N; a.ldc = ldc;
    a.out_scale = 1.f; a.out_scale_dev = out_scale_dev;
    a.ksplit = ksplit; a.part = const_cast<float*>(part);
    const long long quads = (long long)M * (N >> 2);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256), 1, nprob), dim3(256), 0, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_gemm_split_glds(const GemmDesc& d, hipStream_t s);
int gemm_split_glds_shape(const GemmDesc& d);  // 4 = 256x256 blocks, 1 = 128x128
bool gemm_f32_persist_applies(const GemmDesc& d);  // gemm_f32p.hip: exact f32, persistent direct-to-LDS form (bit-identical to the kernels here)
int launch_gemm_f32_persist(const GemmDesc& d, hipStream_t s);

int launch_gemm(const GemmDesc& d, hipStream_t s) {
    SOLA_ARG(d.nprob >= 1 && d.nprob <= 3, "gemm: nprob %d", d.nprob);
    SOLA_ARG(d.M > 0 && d.N > 0 && d.K > 0 && d.K % 4 == 0, "gemm: bad dims M=%d N=%d K=%d (K %% 4 == 0 required)", d.M, d.N, d.K);
    if (d.conv) {
        SOLA_ARG(d.Cin % 4 == 0 && d.K % d.Cin == 0 && d.K / d.Cin <= 8, "conv gemm: Cin=%d K=%d (at most 8 taps)", d.Cin, d.K);
    } else {
        SOLA_ARG(d.lda % 4 == 0, "gemm: lda %d must be a multiple of 4", d.lda);
    }
    GemmArgs a;
    for (int i = 0; i < 3; ++i) a.p[i] = d.p[i < d.nprob ? i : 0];
    a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldr = d.ldr; a.ldc = d.ldc;
    a.conv = d.conv; a.T_in = d.T_in; a.T_out = d.T_out; a.stride = d.stride; a.pad = d.pad; a.Cin = d.Cin;
    a.rowmap = d.conv == 1 ? d.rowmap : nullptr;
    a.tiles_m = a.tiles_n = a.xcd_remap = 0;
    a.out_scale = d.arith >= 1 && d.out_scale != 0.f ? d.out_scale : 1.f;
    a.out_scale_dev = d.arith >= 1 ? d.out_scale_dev : nullptr;
    a.r_sp16 = d.r_sp16;
    a.c_sp16 = d.arith == 1 ? d.c_sp16 : 0;
    a.ksplit = 1; a.kt_per = 0; a.part = nullptr;
    a.guard = a.c_sp16 ? d.guard : nullptr;
    for (int j = 0; j < 3; ++j) a.w_nn[j] = d.w_nn_rows ? d.w_nn[j] : nullptr;
    a.w_nn_rows = d.w_nn_rows;
    SOLA_ARG(!d.w_nn_rows || gemm_nn_supported(d), "gemm: the row-major weight form needs the few-row exact-f32 shape (M=%d N=%d K=%d, rows per matrix %d)", d.M, d.N, d.K, d.w_nn_rows);
    SOLA_ARG(!a.c_sp16 || (d.N % 8 == 0 && d.ldc % 8 == 0), "gemm: split-f16 output needs N %% 8 == 0 and ldc %% 8 == 0");
    const long long t128 = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.nprob;
    const bool big = t128 >= 512;  // two 128x128 blocks per CU x 256 CUs
    if (d.arith == 2) {  // plain f16 operands: direct-to-LDS kernels for every grid size
        SOLA_ARG(gemm_split_glds_supported(d), "f16 gemm: K %% 64, row pitch %% 16 (Cin %% 64 for convs) required; K=%d lda=%d Cin=%d", d.K, d.lda, d.Cin);
        SOLA_ARG(!d.c_f16 || (d.N % 4 == 0 && d.ldc % 4 == 0), "f16 gemm: f16 output needs N %% 4 == 0 and ldc %% 4 == 0");
        SOLA_ARG(!d.p[0].R || !d.r_f16 || d.ldr % 4 == 0, "f16 gemm: f16 residual needs ldr %% 4 == 0");
        SolaProfScope prof(d.ksplit > 1 || gemm_split_glds_shape(d) == 4 ? SOLA_PROF_GEMM_SPLIT256 : SOLA_PROF_GEMM_SPLIT, s, 2.0 * d.M * d.N * (double)d.K * d.nprob,
                           2.0 * d.nprob * ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N));
        if (d.ksplit > 1) {  // weight gradients (gemm_tn_split.hip): K ranges as work items of the persistent kernel + ordered reduce
            SOLA_ARG(d.splitk_ws && d.K % (64 * d.ksplit) == 0 && d.K / 64 / d.ksplit >= 2 && d.N % 4 == 0 && d.ldc % 4 == 0 && !d.c_f16 &&
                         d.splitk_bytes >= (size_t)d.nprob * d.ksplit * d.M * d.N * sizeof(float),
                     "f16 gemm: split-K over %d ranges needs K %% (64 * ksplit) == 0 (K=%d) and a scratch of nprob*ksplit*M*N floats", d.ksplit, d.K);
            a.ksplit = d.ksplit;
            a.part = d.splitk_ws;
            SOLA_TRY(launch_gemm_split_glds(d, s));
            const long long quads = (long long)a.M * (a.N >> 2);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256), 1, d.nprob), dim3(256), 0, s, a);
            SOLA_LAUNCH_CHECK();
            return SOLA_OK;
        }
        return launch_gemm_split_glds(d, s);
    }
    const bool glds = d.arith == 1 && (big || g_gemm_glds_force || d.ksplit > 1) && g_gemm_glds && gemm_split_glds_supported(d);
    const int cat = d.arith == 1 ? (glds && gemm_split_glds_shape(d) == 4 ? (d.gn_gamma ? SOLA_PROF_GEMM_SPLIT256_GN : SOLA_PROF_GEMM_SPLIT256) : SOLA_PROF_GEMM_SPLIT)
                                 : (big ? SOLA_PROF_GEMM : SOLA_PROF_GEMM_SMALL);
    SolaProfScope prof(cat, s, 2.0 * d.M * d.N * (double)d.K * d.nprob,
                       4.0 * d.nprob * ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N));
    if (gemm_small_applies(d)) return launch_small(a, d.nprob, s);
    if (d.arith == 0 && big && gemm_f32_persist_applies(d)) return launch_gemm_f32_persist(d, s);
    const int pipe = g_gemm_variant < 0 ? (big ? 0 : 1) : g_gemm_variant;
    // small grids (fewer 64x64 tiles than CUs, the single-sample regime): split K over up to 8 blocks per tile so the
    // serial k-loop gets ~8x shorter; partial sums go through the caller's scratch and are reduced in a fixed order
    {
        const long long t64 = (long long)((d.M + 63) / 64) * ((d.N + 63) / 64) * d.nprob;
        const int nk_all = (d.K + BK - 1) / BK;
        if (g_gemm_splitk && !big && d.splitk_ws && t64 < g_gemm_splitk_tiles && nk_all >= 8 && d.N % 4 == 0 && d.ldc % 4 == 0 && (!a.c_sp16 || d.N % 8 == 0)) {
            long long S = std::min<long long>(g_gemm_splitk_max, std::min<long long>(nk_all / (g_gemm_splitk_max > 8 ? 2 : 4), (2 * g_gemm_splitk_tiles + t64 - 1) / t64));
            const long long per_split = (long long)d.nprob * d.M * d.N * 4;
            S = std::min<long long>(S, (long long)(d.splitk_bytes / (size_t)per_split));
            if (S >= 2) {
                a.kt_per = (int)((nk_all + S - 1) / S);
                a.ksplit = (nk_all + a.kt_per - 1) / a.kt_per;  // no empty split
                a.part = d.splitk_ws;
            }
        }
    }
    if (glds && d.ksplit > 1) {
        SOLA_ARG(d.splitk_ws && d.K % (32 * d.ksplit) == 0 && d.K / 32 / d.ksplit >= 2 && d.N % 4 == 0 && d.ldc % 4 == 0 &&
                     d.splitk_bytes >= (size_t)d.nprob * d.ksplit * d.M * d.N * sizeof(float),
                 "gemm: split-K over %d ranges needs K %% (32 * ksplit) == 0 (K=%d) and a scratch of nprob*ksplit*M*N floats", d.ksplit, d.K);
        a.ksplit = d.ksplit;
        a.part = d.splitk_ws;
        SOLA_TRY(launch_gemm_split_glds(d, s));
        const long long quads = (long long)a.M * (a.N >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256), 1, d.nprob), dim3(256), 0, s, a);
        SOLA_LAUNCH_CHECK();
        return SOLA_OK;
    }
    if (glds) return launch_gemm_split_glds(d, s);
    SOLA_ARG(!d.gn_gamma, "gemm: the fused GroupNorm epilogue needs the persistent direct-to-LDS kernel (check gemm_gn_fusable)");
    if (d.arith == 1) {
        SOLA_ARG(d.K % 16 == 0 && (d.conv ? d.Cin % 8 == 0 : d.lda % 8 == 0), "split-f16 gemm: K %% 16 and row pitch %% 8 required");
        if (big) return pipe ? launch_tile<128, 128, 1, 1>(a, d.nprob, s) : launch_tile<128, 128, 0, 1>(a, d.nprob, s);
        return pipe ? launch_tile<64, 64, 1, 1>(a, d.nprob, s) : launch_tile<64, 64, 0, 1>(a, d.nprob, s);
    }
    if (big && g_gemm_f32_nw8 && !pipe) return launch_tile<128, 128, 0, 0, 8>(a, d.nprob, s);  // (the mid-tile staging schedule - an A/B - keeps four waves: at eight it spills)
    if (big) return pipe ? launch_tile<128, 128, 1, 0>(a, d.nprob, s) : launch_tile<128, 128, 0, 0>(a, d.nprob, s);
    return pipe ? launch_tile<64, 64, 1, 0>(a, d.nprob, s) : launch_tile<64, 64, 0, 0>(a, d.nprob, s);
}
