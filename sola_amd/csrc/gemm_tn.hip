// Weight-gradient GEMM and its helpers (backward of F.linear / F.conv1d at tools/attention.py:63-73, module/ws.py:14-22):
//   gemm_tn_f32_kernel : P[s][n][k] = sum_{m in split s} A[m][n] * B[m][k]   ("TN": both operands reduce over ROWS)
//                        A = dY [M, N]; B = X [M, K] or the implicit im2col of the channels-last conv input.
//   sum_slabs_kernel   : C[n][k] = sum_s P[s][n][k]  (fixed order: deterministic, no float atomics)
//   transpose_kernel   : out[c][r] = in[r][c]  (W -> W^T so that dX = dY * W runs on the NT kernel)
//   colsum_kernel      : out[seg][c] = sum_{r in segment} in[r][c]  (bias gradients, GroupNorm dgamma/dbeta, dlbar)
// Exact f32 on v_mfma_f32_32x32x2_f32.  The reduction index m is the slow (row) index of both operands, so tiles are
// staged [32 m][128 cols] with coalesced float4 row loads and the MFMA fragments are ds_read_b32 across the columns
// (lanes 0-31 take row m, lanes 32-63 row m+1: consecutive addresses, conflict-free).
#include <algorithm>
#include <utility>

#include "kernels.h"

namespace {

struct TnArgs {
    const float* A;  // [M, N]
    const float* B;  // [M, K] or conv source
    float* P;        // [splits][N][K]
    float* Pb;       // optional [splits][N]: column sums of A (bias gradient), written by the tk == 0 blocks
    int M, N, K, lda, ldb;
    int conv, T_in, T_out, stride, pad, Cin;  // conv gather on B (same convention as the NT kernel's mode 1)
    const int2* rowmap;  // conv, ragged batches: row m reads tap kk from source row rowmap[m].x + kk when bit kk of rowmap[m].y is set
    int tiles_n, tiles_k, m_per_split;
};

constexpr int TB = 128;       // output tile (n and k)
constexpr int TM = 32;        // rows of the reduction per LDS stage
constexpr int TP = TB + 4;    // LDS pitch (floats), 16-byte aligned rows

// one (tile, row split) work item: tile = tn * tiles_k + tk
// NW = 4: 2 x 2 waves of 64 x 64; NW = 8 (round 4): 4 x 2 waves of 32 x 64 - the CU's two blocks (LDS) put four waves on every SIMD instead of
// two, as in the NT kernel (gemm.hip); the same reduction order per output element: bit-identical.
template <int NW>
__device__ __forceinline__ void tn_tile(const TnArgs& a, int tile, int split, float* smem) {
    constexpr int TMW = NW == 8 ? 1 : 2;     // 32-row MFMA tiles per wave along n
    constexpr int LP = NW == 8 ? 2 : 4;      // load passes: 8 rows per pass at 256 threads, 16 at 512
    constexpr int LR = NW == 8 ? 16 : 8;
    float* As = smem;                 // [2][TM][TP]
    float* Bs = smem + 2 * TM * TP;   // [2][TM][TP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tn = tile / a.tiles_k, tk = tile % a.tiles_k;
    const int n0 = tn * TB, k0 = tk * TB;
    const int m_begin = split * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);

    // loads: 32 lanes cover one 128-float row; LR rows per pass, LP passes
    const int lrow = tid >> 5;         // 0..LR-1
    const int lcol = (tid & 31) << 2;  // 0..124
    float4 ra[LP], rb[LP];
    auto load_stage = [&](int mbase) {
        int kk = 0, ci = k0 + lcol;
        if (a.conv) {
            kk = ci / a.Cin;
            ci -= kk * a.Cin;
        }
#pragma unroll
        for (int i = 0; i < LP; ++i) {
            const int m = mbase + lrow + LR * i;
            const bool mok = m < m_end;
            ra[i] = (mok && n0 + lcol < a.N) ? *reinterpret_cast<const float4*>(a.A + (long long)m * a.lda + n0 + lcol)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = mok && k0 + lcol < a.K;
            const float* src;
            if (a.conv && a.rowmap) {
                const int2 rm = a.rowmap[mok ? m : 0];
                ok = ok && ((rm.y >> kk) & 1);
                src = a.B + ((long long)rm.x + kk) * a.Cin + ci;
            } else if (a.conv) {
                const int r = m / a.T_out;
                const int to = m - r * a.T_out;
                const int ti = to * a.stride - a.pad + kk;
                ok = ok && ti >= 0 && ti < a.T_in;
                src = a.B + ((long long)r * a.T_in + ti) * a.Cin + ci;
            } else {
                src = a.B + (long long)m * a.ldb + k0 + lcol;
            }
            rb[i] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LP; ++i) {
            *reinterpret_cast<float4*>(&As[(buf * TM + lrow + LR * i) * TP + lcol]) = ra[i];
            *reinterpret_cast<float4*>(&Bs[(buf * TM + lrow + LR * i) * TP + lcol]) = rb[i];
        }
    };

    f32x16 acc[TMW][2];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool want_bias = a.Pb != nullptr && tk == 0;

    const int fcol = lane & 31, fm = lane >> 5;
    const int nstage = (m_end - m_begin + TM - 1) / TM;
    if (nstage > 0) {
        load_stage(m_begin);
        if (want_bias) {
#pragma unroll
            for (int i = 0; i < LP; ++i) { bsum.x += ra[i].x; bsum.y += ra[i].y; bsum.z += ra[i].z; bsum.w += ra[i].w; }
        }
        store_stage(0);
    }
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        if (st + 1 < nstage) load_stage(m_begin + (st + 1) * TM);
        const float* Ab = &As[(buf * TM + fm) * TP + wr * (32 * TMW) + fcol];
        const float* Bb = &Bs[(buf * TM + fm) * TP + wc * 64 + fcol];
        // fragments of four row pairs at a time, the next four requested before the current four's MFMAs (as one unrolled loop the compiler
        // issued read -> full wait -> two MFMAs per row pair: the LDS latency sat between every pair of MFMAs)
        constexpr int CHK = 4, NCH = TM / 2 / CHK;
        float fa0[2][CHK], fa1[2][CHK], fb0[2][CHK], fb1[2][CHK];
        auto frag = [&](int cidx, int slot) {
#pragma unroll
            for (int e = 0; e < CHK; ++e) {
                const int mm = (cidx * CHK + e) * 2;
                fa0[slot][e] = Ab[mm * TP];
                if constexpr (TMW == 2) fa1[slot][e] = Ab[mm * TP + 32];
                fb0[slot][e] = Bb[mm * TP];
                fb1[slot][e] = Bb[mm * TP + 32];
            }
        };
        frag(0, 0);
#pragma unroll
        for (int cidx = 0; cidx < NCH; ++cidx) {
            const int slot = cidx & 1;
            if (cidx + 1 < NCH) frag(cidx + 1, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);  // keep the next chunk's reads in front of this chunk's MFMAs (the scheduler sinks loads to their uses)
#pragma unroll
            for (int e = 0; e < CHK; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[slot][e], fb0[slot][e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[slot][e], fb1[slot][e], acc[0][1], 0, 0, 0);
                if constexpr (TMW == 2) {
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[slot][e], fb0[slot][e], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[slot][e], fb1[slot][e], acc[1][1], 0, 0, 0);
                }
            }
        }
        if (st + 1 < nstage) {
            if (want_bias) {
#pragma unroll
                for (int i = 0; i < LP; ++i) { bsum.x += ra[i].x; bsum.y += ra[i].y; bsum.z += ra[i].z; bsum.w += ra[i].w; }
            }
            store_stage(buf ^ 1);
        }
        __syncthreads();
    }
    if (want_bias) {
        float* red = smem;  // [LR][TB]: reduce the LR row lanes (the stage buffers are free after the last barrier)
        *reinterpret_cast<float4*>(&red[lrow * TB + lcol]) = bsum;
        __syncthreads();
        if (tid < TB && n0 + tid < a.N) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < LR; ++r) v += red[r * TB + tid];
            a.Pb[(long long)split * a.N + n0 + tid] = v;
        }
    }
    float* P = a.P + (long long)split * a.N * a.K;
    const int col_l = lane & 31, row_l = (lane >> 5) << 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + wc * 64 + j * 32 + col_l;
        if (k >= a.K) continue;
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * (32 * TMW) + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
                if (n < a.N) P[(long long)n * a.K + k] = acc[i][j][r];
            }
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void gemm_tn_f32_kernel(const TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    tn_tile<NW>(a, (int)blockIdx.x, (int)blockIdx.y, smem);
}

// ---- grouped form (round 4): the weight gradients of ALL linear layers of a few-sample step in ONE launch --------------------------
// One sample per optimizer step (the reference's batch size) has 256-2000 token rows: a [1024 x 1024] weight gradient is 64 tiles
// whose reduction over the rows is 8-60 stages.  Per matrix the launch above cut the rows into slabs to fill the chip (16 MB of
// partial sums written and read back per matrix) and two sum_slabs launches folded them: 24 + 48 launches of 5-17 us per step, 0.63
// ms of a 2.7 ms step.  The backward now defers these products (backward.hip): each dY keeps a buffer of its own until the end, and
// this kernel takes them all - blockIdx.x = (problem, tile), every block reduces over ALL rows of its problem and writes its dW tile
// and bias gradient directly (no partial sums, no second pass, fixed summation order).
constexpr int TN_GROUP_MAX = 32;
struct TnGroupProb {
    const float* A;
    const float* B;
    float* C;
    float* db;
    const int2* rowmap;
    int M, N, K, lda, ldb;
    int conv, T_in, T_out, stride, pad, Cin;
    int tiles_k, tile_begin;  // this problem's tiles are [tile_begin, next problem's tile_begin)
};
struct TnGroupArgs {
    TnGroupProb p[TN_GROUP_MAX];
    int nprob;
};
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void gemm_tn_f32_group_kernel(const TnGroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int pi = 0;
    for (int j = 1; j < g.nprob; ++j)  // block-uniform, at most 32 entries
        if ((int)blockIdx.x >= g.p[j].tile_begin) pi = j;
    const TnGroupProb& q = g.p[pi];
    TnArgs a;
    a.A = q.A; a.B = q.B; a.P = q.C; a.Pb = q.db;
    a.M = q.M; a.N = q.N; a.K = q.K; a.lda = q.lda; a.ldb = q.ldb;
    a.conv = q.conv; a.T_in = q.T_in; a.T_out = q.T_out; a.stride = q.stride; a.pad = q.pad; a.Cin = q.Cin; a.rowmap = q.rowmap;
    a.tiles_n = 0; a.tiles_k = q.tiles_k;
    a.m_per_split = (a.M + TM - 1) / TM * TM;  // one split: every row
    tn_tile<NW>(a, (int)blockIdx.x - q.tile_begin, 0, smem);
}

// (round 4's grouped weight transposition - W -> W^T for the dX GEMMs of a few-sample backward, 49 matrices in one launch - left with round 5:
// those GEMMs read the weights where they lie, gemm.hip's NN form)

__global__ void sum_slabs_kernel(const float* __restrict__ P, float* __restrict__ C, long long n4, int splits) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 acc = reinterpret_cast<const float4*>(P)[i];
    for (int s = 1; s < splits; ++s) {
        const float4 v = reinterpret_cast<const float4*>(P)[i + (long long)s * n4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(C)[i] = acc;
}

// out[c * ldo + col_off + r] = in[r * ldi + c], 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                        int cols, int ldi, int ldo, int col_off) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * i][tx] = in[(long long)r * ldi + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < rows && c < cols) out[(long long)c * ldo + col_off + r] = tile[tx][ty + 8 * i];
    }
}

// The same transposition written straight in a GEMM operand format (FMT 0 = split-f16 pairs in 8-value blocks [hi8 | lo8], 1 = f16,
// 2 = bfloat16), scaled: out row c (pitch ldo VALUES) column col_off + r = scale * in[r][c].  One launch where the backward's dX GEMMs
// ran a transposition into an f32 buffer and a cast of that buffer (two launches per weight matrix and step).
template <int FMT>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ in, void* __restrict__ out, int rows, int cols, int ldi,
                                                             int ldo, int col_off, float scale) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * i][tx] = in[(long long)r * ldi + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r >= rows || c >= cols) continue;
        const float v = tile[tx][ty + 8 * i] * scale;
        const long long n = col_off + r;
        if (FMT == 0) {
            _Float16 hi, lo;
            split_f16(v, hi, lo);
            _Float16* blk = reinterpret_cast<_Float16*>(static_cast<float*>(out) + (long long)c * ldo + (n & ~7LL)) + (n & 7);
            blk[0] = hi;
            blk[8] = lo;
        } else if (FMT == 1) {
            static_cast<_Float16*>(out)[(long long)c * ldo + n] = (_Float16)v;
        } else {
            static_cast<__bf16*>(out)[(long long)c * ldo + n] = (__bf16)v;
        }
    }
}

// out[(chunk * segments + seg)][c] (+)= scale * sum_{r in chunk of segment} in[(seg * seg_rows + r)][c]
// block = 64 columns x 4 row lanes; grid.z splits long segments into row chunks (second launch adds the chunks up)
// seg_stride: elements between the first rows of consecutive segments (seg_rows * ld for back-to-back segments).  out1 / split: output
// index >= split goes to out1[index - split] instead (two reductions with separate destinations in one launch: launch_colsum_pair).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, float* __restrict__ out, int seg_rows,
                                                     int cols, int ld, int chunk_rows, float scale, int accumulate, long long seg_stride,
                                                     float* __restrict__ out1, long long split) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int seg = blockIdx.y;
    const int r_begin = blockIdx.z * chunk_rows;
    const int r_end = min(seg_rows, r_begin + chunk_rows);
    // four independent partial sums, fixed combination order: a thread's loads pipeline instead of queueing behind one add
    // chain (a [256][1024] bias pass took 16 us for 1 MB: 64 dependent load-add steps per thread)
    // (round 3: eight - the 70 second-stage bias / norm-parameter reductions of a training step are ~200 dependent rows each)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    if (c < cols) {
        const float* p = in + (long long)seg * seg_stride + c;
        int r = r_begin + rl;
        for (; r + 28 < r_end; r += 32) {
            s0 += p[(long long)r * ld]; s1 += p[(long long)(r + 4) * ld]; s2 += p[(long long)(r + 8) * ld]; s3 += p[(long long)(r + 12) * ld];
            s4 += p[(long long)(r + 16) * ld]; s5 += p[(long long)(r + 20) * ld]; s6 += p[(long long)(r + 24) * ld]; s7 += p[(long long)(r + 28) * ld];
        }
        for (; r < r_end; r += 4) s0 += p[(long long)r * ld];
    }
    const float s = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < cols) {
        const float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) * scale;
        const long long oi = ((long long)blockIdx.z * gridDim.y + seg) * cols + c;
        float* o = (out1 && oi >= split) ? out1 + (oi - split) : out + oi;
        *o = accumulate ? *o + v : v;
    }
}


// up to 16 (dgamma, dbeta) pairs of column sums in ONE launch (round 5, few-sample backward: the 11 GroupNorm backwards of a one-sample step
// each ended in a launch_colsum_pair of ~5 us; deferred - every norm keeps private partial buffers - they are one launch per bucket).  Same
// per-column arithmetic as colsum_kernel with one chunk (rows r = lane, lane + 4, ... in eight partial sums, the four row lanes combined in
// a fixed order): the same bits as launch_colsum_pair on <= 256 rows.
constexpr int CSG_MAX = 16;
struct ColsumGroupArgs {
    const float* in0[CSG_MAX];
    const float* in1[CSG_MAX];
    float* out0[CSG_MAX];
    float* out1[CSG_MAX];
    int rows[CSG_MAX], cols[CSG_MAX], first_block[CSG_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void colsum_pair_group_kernel(const ColsumGroupArgs a) {
    __shared__ float red[4][64];
    int e = 0;
    for (int j = 1; j < a.n; ++j)
        if ((int)blockIdx.x >= a.first_block[j]) e = j;
    const int lb = (int)blockIdx.x - a.first_block[e];
    const int cb = (a.cols[e] + 63) / 64;
    const int which = lb / cb;
    const int c = (lb - which * cb) * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int cols = a.cols[e], ld = cols, r_end = a.rows[e];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    if (c < cols) {
        const float* p = (which ? a.in1[e] : a.in0[e]) + c;
        int r = rl;
        for (; r + 28 < r_end; r += 32) {
            s0 += p[(long long)r * ld]; s1 += p[(long long)(r + 4) * ld]; s2 += p[(long long)(r + 8) * ld]; s3 += p[(long long)(r + 12) * ld];
            s4 += p[(long long)(r + 16) * ld]; s5 += p[(long long)(r + 20) * ld]; s6 += p[(long long)(r + 24) * ld]; s7 += p[(long long)(r + 28) * ld];
        }
        for (; r < r_end; r += 4) s0 += p[(long long)r * ld];
    }
    const float s = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < cols) {
        const float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) * 1.0f;
        (which ? a.out1[e] : a.out0[e])[c] = v;
    }
}

// Up to 48 single column sums in ONE launch (round 6: the ~30 bias gradients of a reduced-precision training step were one ~9-us launch each
// over the slab sums their statistics pass left; every statistics pass keeps its slab table now and a bucket's bias sums leave together).
// Per-column arithmetic = colsum_kernel with one chunk (eight partial sums per row lane, four row lanes combined in a fixed order): the same bits.
constexpr int CSJ_MAX = 48;
struct ColsumJobsArgs {
    const float* in[CSJ_MAX];
    float* out[CSJ_MAX];
    int rows[CSJ_MAX], cols[CSJ_MAX], ld[CSJ_MAX], first_block[CSJ_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void colsum_jobs_kernel(const ColsumJobsArgs a) {
    __shared__ float red[4][64];
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int e = lo;
    const int c = ((int)blockIdx.x - a.first_block[e]) * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int cols = a.cols[e], ld = a.ld[e], r_end = a.rows[e];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    if (c < cols) {
        const float* p = a.in[e] + c;
        int r = rl;
        for (; r + 28 < r_end; r += 32) {
            s0 += p[(long long)r * ld]; s1 += p[(long long)(r + 4) * ld]; s2 += p[(long long)(r + 8) * ld]; s3 += p[(long long)(r + 12) * ld];
            s4 += p[(long long)(r + 16) * ld]; s5 += p[(long long)(r + 20) * ld]; s6 += p[(long long)(r + 24) * ld]; s7 += p[(long long)(r + 28) * ld];
        }
        for (; r < r_end; r += 4) s0 += p[(long long)r * ld];
    }
    const float s = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < cols) a.out[e][c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) * 1.0f;
}

}  // namespace

int launch_colsum_jobs(const ColsumJobsDesc& d, hipStream_t s) {
    SOLA_ARG(d.n >= 1 && d.n <= CSJ_MAX, "colsum_jobs: %d jobs", d.n);
    ColsumJobsArgs a;
    a.n = d.n;
    int blocks = 0;
    double bytes = 0;
    for (int e = 0; e < d.n; ++e) {
        SOLA_ARG(d.in[e] && d.out[e] && d.rows[e] > 0 && d.cols[e] > 0 && d.ld[e] >= d.cols[e], "colsum_jobs: job %d", e);
        a.in[e] = d.in[e]; a.out[e] = d.out[e]; a.rows[e] = d.rows[e]; a.cols[e] = d.cols[e]; a.ld[e] = d.ld[e];
        a.first_block[e] = blocks;
        blocks += (d.cols[e] + 63) / 64;
        bytes += 4.0 * d.rows[e] * d.cols[e];
    }
    a.first_block[d.n] = blocks;
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, bytes);
    hipLaunchKernelGGL(colsum_jobs_kernel, dim3(blocks), dim3(256), 0, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int g_gemm_tn_nw8 = 1;  // sola_tune "gemm_tn_nw8": the exact-f32 weight-gradient kernels with eight waves per block (0 = four; A/B)
void sola_gemm_tn_set_nw8(int v) { g_gemm_tn_nw8 = v; }
static int tn_splits(int M, int N, int K) {
    const long long tiles = (long long)((N + TB - 1) / TB) * ((K + TB - 1) / TB);
    int splits = (int)((768 + tiles - 1) / tiles);
    const int max_splits = (M + 4 * TM - 1) / (4 * TM);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    return splits;
}

int gemm_tn_persist_plan(int M, int N, int K);  // gemm_tn_f32p.hip
size_t colsum_scratch_bytes(int segments, int seg_rows, int cols);
size_t gemm_tn_scratch_bytes(int M, int N, int K) {
    const int splits = tn_splits(M, N, K);
    const size_t one_tile = (size_t)splits * ((size_t)N * K + N) * sizeof(float);
    const int ps = gemm_tn_persist_plan(M, N, K);  // the persistent kernel's partial sums + the bias gradient's column-sum pass behind them
    const size_t persist = ps > 0 ? (size_t)ps * N * K * sizeof(float) + colsum_scratch_bytes(1, M, N) : 0;
    return std::max(one_tile, persist);
}

int gemm_tn_persist_splits(const GemmTnDesc& d, int max_splits);  // gemm_tn_f32p.hip: the persistent direct-to-LDS form (0 = not its shape)
int launch_gemm_tn_f32_persist(const GemmTnDesc& d, int splits, float* P, int* splits_out, hipStream_t s);

int launch_gemm_tn(const GemmTnDesc& d, hipStream_t s) {
    SOLA_ARG(d.M > 0 && d.N > 0 && d.K > 0 && d.N % 4 == 0 && d.K % 4 == 0, "gemm_tn: bad dims M=%d N=%d K=%d", d.M, d.N, d.K);
    SOLA_ARG(d.lda % 4 == 0 && (d.conv || d.ldb % 4 == 0), "gemm_tn: leading dims must be multiples of 4");
    if (d.conv) SOLA_ARG(d.Cin % 4 == 0 && d.K % d.Cin == 0, "gemm_tn conv: Cin=%d K=%d", d.Cin, d.K);
    TnArgs a;
    a.A = d.A; a.B = d.B; a.P = d.scratch; a.Pb = nullptr; a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldb = d.ldb;
    a.conv = d.conv; a.T_in = d.T_in; a.T_out = d.T_out; a.stride = d.stride; a.pad = d.pad; a.Cin = d.Cin;
    a.rowmap = d.conv ? d.rowmap : nullptr;
    a.tiles_n = (d.N + TB - 1) / TB;
    a.tiles_k = (d.K + TB - 1) / TB;
    const size_t need = gemm_tn_scratch_bytes(d.M, d.N, d.K);
    if (d.scratch_bytes < need) {
        sola_set_error("gemm_tn: scratch %zu < %zu", d.scratch_bytes, need);
        return SOLA_ERR_WORKSPACE;
    }
    const int splits = tn_splits(d.M, d.N, d.K);
    // round 5: the persistent kernel where the shape is its own (N % 256, K % 128, a grid that fills whole rounds).  Its partial sums stay
    // inside the scratch sized for `splits`; the bias gradient (column sums of dY) is a pass of its own there - launch_colsum in the
    // slab a split less leaves free (the one-tile kernel sums the rows it stages, in registers: vector adds the f32 MFMA cannot hide)
    {
        const size_t cs_need = d.bias_grad ? colsum_scratch_bytes(1, d.M, d.N) : 0;
        const int room = (int)std::min<size_t>(64, (d.scratch_bytes - std::min(d.scratch_bytes, cs_need)) / ((size_t)d.N * d.K * sizeof(float)));
        const int ps = room >= 1 ? gemm_tn_persist_splits(d, room) : 0;
        if (ps > 0) {
            int used = 0;
            {
                SolaProfScope prof(SOLA_PROF_GEMM_TN, s, 2.0 * d.M * d.N * (double)d.K, 4.0 * ((double)d.M * (d.N + d.K) + (double)ps * d.N * d.K));
                SOLA_TRY(launch_gemm_tn_f32_persist(d, ps, d.scratch, &used, s));
            }
            const long long n4 = (long long)d.N * d.K / 4;
            {
                SolaProfScope prof(SOLA_PROF_MISC, s, 0, 4.0 * (used + 1.0) * d.N * d.K);
                hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, d.scratch, d.C, n4, used);
                SOLA_LAUNCH_CHECK();
            }
            if (d.bias_grad) {
                float* cs = d.scratch + (size_t)used * d.N * d.K;
                SOLA_TRY(launch_colsum(d.A, d.bias_grad, 1, d.M, d.N, d.lda, 1.0f, 0, cs, d.scratch_bytes - (size_t)used * d.N * d.K * sizeof(float), s));
            }
            return SOLA_OK;
        }
    }
    if (d.bias_grad) a.Pb = d.scratch + (size_t)splits * d.N * d.K;
    int mps = (d.M + splits - 1) / splits;
    mps = (mps + TM - 1) / TM * TM;
    a.m_per_split = mps;
    constexpr size_t lds = (size_t)4 * TM * TP * sizeof(float);
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_f32_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_f32_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.done(dev);
    }
    {
        SolaProfScope prof(SOLA_PROF_GEMM_TN, s, 2.0 * d.M * d.N * (double)d.K, 4.0 * ((double)d.M * (d.N + d.K) + (double)splits * d.N * d.K));
        if (g_gemm_tn_nw8) hipLaunchKernelGGL(gemm_tn_f32_kernel<8>, dim3(a.tiles_n * a.tiles_k, splits), dim3(512), lds, s, a);
        else hipLaunchKernelGGL(gemm_tn_f32_kernel<4>, dim3(a.tiles_n * a.tiles_k, splits), dim3(256), lds, s, a);
        SOLA_LAUNCH_CHECK();
    }
    {
        const long long n4 = (long long)d.N * d.K / 4;
        SolaProfScope prof(SOLA_PROF_MISC, s, 0, 4.0 * (splits + 1.0) * d.N * d.K);
        hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, d.scratch, d.C, n4, splits);
        SOLA_LAUNCH_CHECK();
        if (d.bias_grad) {
            const long long b4 = d.N / 4;
            hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((b4 + 255) / 256)), dim3(256), 0, s, a.Pb, d.bias_grad, b4, splits);
            SOLA_LAUNCH_CHECK();
        }
    }
    return SOLA_OK;
}

int launch_gemm_tn_group(const GemmTnGroupDesc& d, hipStream_t s) {
    SOLA_ARG(d.nprob >= 1 && d.nprob <= TN_GROUP_MAX, "gemm_tn_group: nprob=%d", d.nprob);
    TnGroupArgs g;
    g.nprob = d.nprob;
    double flops = 0, bytes = 0;
    int tiles = 0;
    for (int j = 0; j < d.nprob; ++j) {
        const GemmTnGroupDesc::Prob& q = d.p[j];
        SOLA_ARG(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K > 0 && q.N % 4 == 0 && q.K % 4 == 0 && q.lda % 4 == 0 && (q.conv || q.ldb % 4 == 0),
                 "gemm_tn_group: problem %d (M=%d N=%d K=%d)", j, q.M, q.N, q.K);
        if (q.conv) SOLA_ARG(q.Cin % 4 == 0 && q.K % q.Cin == 0, "gemm_tn_group conv: Cin=%d K=%d", q.Cin, q.K);
        TnGroupProb& o = g.p[j];
        o.A = q.A; o.B = q.B; o.C = q.C; o.db = q.bias_grad; o.rowmap = q.conv ? q.rowmap : nullptr;
        o.M = q.M; o.N = q.N; o.K = q.K; o.lda = q.lda; o.ldb = q.ldb;
        o.conv = q.conv; o.T_in = q.T_in; o.T_out = q.T_out; o.stride = q.stride; o.pad = q.pad; o.Cin = q.Cin;
        o.tiles_k = (q.K + TB - 1) / TB;
        o.tile_begin = tiles;
        tiles += ((q.N + TB - 1) / TB) * o.tiles_k;
        flops += 2.0 * q.M * q.N * (double)q.K;
        bytes += 4.0 * ((double)q.M * (q.N + q.K) + (double)q.N * q.K);
    }
    constexpr size_t lds = (size_t)4 * TM * TP * sizeof(float);
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_f32_group_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_f32_group_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.done(dev);
    }
    SolaProfScope prof(SOLA_PROF_GEMM_TN, s, flops, bytes);
    if (g_gemm_tn_nw8) hipLaunchKernelGGL(gemm_tn_f32_group_kernel<8>, dim3((unsigned)tiles), dim3(512), lds, s, g);
    else hipLaunchKernelGGL(gemm_tn_f32_group_kernel<4>, dim3((unsigned)tiles), dim3(256), lds, s, g);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_transpose(const float* in, float* out, int rows, int cols, int ldi, int ldo, int col_off, hipStream_t s) {
    SOLA_ARG(rows > 0 && cols > 0, "transpose: bad dims");
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 8.0 * rows * cols);
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, in, out, rows, cols, ldi, ldo, col_off);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_transpose_cast(const float* in, void* out, int rows, int cols, int ldi, int ldo, int col_off, float scale, int fmt, hipStream_t s) {
    SOLA_ARG(in && out && rows > 0 && cols > 0 && fmt >= 0 && fmt <= 2 && (fmt != 0 || ldo % 8 == 0), "transpose_cast: bad arguments (fmt %d ldo %d)", fmt, ldo);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, (fmt == 0 ? 8.0 : 6.0) * rows * cols);
    const dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    if (fmt == 0) hipLaunchKernelGGL(transpose_cast_kernel<0>, grid, dim3(256), 0, s, in, out, rows, cols, ldi, ldo, col_off, scale);
    else if (fmt == 1) hipLaunchKernelGGL(transpose_cast_kernel<1>, grid, dim3(256), 0, s, in, out, rows, cols, ldi, ldo, col_off, scale);
    else hipLaunchKernelGGL(transpose_cast_kernel<2>, grid, dim3(256), 0, s, in, out, rows, cols, ldi, ldo, col_off, scale);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

size_t colsum_scratch_bytes(int segments, int seg_rows, int cols) {
    if (seg_rows <= 256) return 0;
    const int chunks = min(64, (seg_rows + 127) / 128);
    return (size_t)chunks * segments * cols * sizeof(float);
}

int launch_colsum(const float* in, float* out, int segments, int seg_rows, int cols, int ld, float scale, int accumulate,
                  float* scratch, size_t scratch_bytes, hipStream_t s) {
    SOLA_ARG(segments > 0 && seg_rows > 0 && cols > 0 && segments <= 65535, "colsum: bad dims");
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 4.0 * segments * (double)seg_rows * cols);
    const size_t need = colsum_scratch_bytes(segments, seg_rows, cols);
    if (need == 0 || scratch == nullptr || scratch_bytes < need) {
        hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, segments, 1), dim3(256), 0, s, in, out, seg_rows, cols, ld,
                           seg_rows, scale, accumulate, (long long)seg_rows * ld, (float*)nullptr, 0LL);
        SOLA_LAUNCH_CHECK();
        return SOLA_OK;
    }
    const int chunks = (int)(need / ((size_t)segments * cols * sizeof(float)));
    const int chunk_rows = (seg_rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, segments, chunks), dim3(256), 0, s, in, scratch, seg_rows, cols,
                       ld, chunk_rows, 1.0f, 0, (long long)seg_rows * ld, (float*)nullptr, 0LL);
    SOLA_LAUNCH_CHECK();
    const int wide = segments * cols;  // second pass: [chunks][segments*cols] -> [segments*cols]
    hipLaunchKernelGGL(colsum_kernel, dim3((wide + 63) / 64, 1, 1), dim3(256), 0, s, scratch, out, chunks, wide, wide, chunks,
                       scale, accumulate, (long long)chunks * wide, (float*)nullptr, 0LL);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

// Two column sums of equally shaped matrices with separate destinations (GroupNorm's dgamma / dbeta partials) in the launches of one:
// the same chunking and summation order as two launch_colsum calls, so the same bits.
int launch_colsum_pair(const float* in0, const float* in1, float* out0, float* out1, int seg_rows, int cols, int ld, float* scratch,
                       size_t scratch_bytes, hipStream_t s) {
    SOLA_ARG(in0 && in1 && out0 && out1 && seg_rows > 0 && cols > 0, "colsum_pair: bad arguments");
    if (in1 < in0) { std::swap(in0, in1); std::swap(out0, out1); }
    const long long stride = in1 - in0;
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 8.0 * (double)seg_rows * cols);
    const size_t need = colsum_scratch_bytes(2, seg_rows, cols);
    if (need == 0 || scratch == nullptr || scratch_bytes < need) {
        hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, 2, 1), dim3(256), 0, s, in0, out0, seg_rows, cols, ld, seg_rows, 1.0f, 0, stride, out1,
                           (long long)cols);
        SOLA_LAUNCH_CHECK();
        return SOLA_OK;
    }
    const int chunks = (int)(need / ((size_t)2 * cols * sizeof(float)));
    const int chunk_rows = (seg_rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, 2, chunks), dim3(256), 0, s, in0, scratch, seg_rows, cols, ld, chunk_rows, 1.0f, 0, stride,
                       (float*)nullptr, 0LL);
    SOLA_LAUNCH_CHECK();
    const int wide = 2 * cols;
    hipLaunchKernelGGL(colsum_kernel, dim3((wide + 63) / 64, 1, 1), dim3(256), 0, s, scratch, out0, chunks, wide, wide, chunks, 1.0f, 0,
                       (long long)chunks * wide, out1, (long long)cols);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}


int launch_colsum_pair_group(const ColsumPairGroupDesc& d, hipStream_t s) {
    SOLA_ARG(d.n >= 1 && d.n <= CSG_MAX, "colsum_pair_group: %d pairs", d.n);
    ColsumGroupArgs a;
    a.n = d.n;
    int blocks = 0;
    double bytes = 0;
    for (int e = 0; e < d.n; ++e) {
        SOLA_ARG(d.in0[e] && d.in1[e] && d.out0[e] && d.out1[e] && d.rows[e] > 0 && d.rows[e] <= 256 && d.cols[e] > 0, "colsum_pair_group: pair %d (rows %d)", e, d.rows[e]);
        a.in0[e] = d.in0[e]; a.in1[e] = d.in1[e]; a.out0[e] = d.out0[e]; a.out1[e] = d.out1[e]; a.rows[e] = d.rows[e]; a.cols[e] = d.cols[e];
        a.first_block[e] = blocks;
        blocks += 2 * ((d.cols[e] + 63) / 64);
        bytes += 8.0 * d.rows[e] * d.cols[e];
    }
    a.first_block[d.n] = blocks;
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, bytes);
    hipLaunchKernelGGL(colsum_pair_group_kernel, dim3(blocks), dim3(256), 0, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
