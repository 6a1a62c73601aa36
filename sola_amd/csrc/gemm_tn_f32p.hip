// Exact-f32 weight-gradient GEMM, persistent direct-to-LDS form (round 5): P[s][n][k] = sum over the rows m of split s of A[m][n] * B[m][k]
// (A = dY [M, N]; B = X [M, K] or the implicit im2col of the channels-last conv input) - the backward of F.linear / F.conv1d at
// tools/attention.py:63-73, module/ws.py:14-22.  gemm_tn.hip's sum_slabs_kernel folds the splits in index order (deterministic).
//
// Same arithmetic as gemm_tn_f32_kernel (v_mfma_f32_32x32x2_f32, one accumulator per output element, rows in ascending pairs inside a
// split); built like gemm_f32p.hip's NT kernel, for the same reason - on gfx950 a VALU instruction does not overlap with the f32 MFMA
// (profiles/r05_mfma_f32_valu.txt), and the one-tile kernel staged global -> VGPR -> LDS with per-row 64-bit address arithmetic:
//   * one block of eight waves per CU walks work items (256 x 128 output tile, row split); wave tile 64 x 64;
//   * a stage = 32 rows of both operands, brought to LDS by buffer loads (descriptor of the split's rows: rows past its end return
//     zeros; a per-lane offset that never changes; the row and column window as a scalar offset).  The reduction runs over the ROWS, so
//     the tile is stored as it lies in memory ([32 m][256 n] and [32 m][128 k], no swizzle) and a fragment is one float per lane read
//     across the columns (lanes 0-31 row 2t, lanes 32-63 row 2t + 1: conflict-free halves), two row pairs per ds_read2st64_b32;
//   * two stages, one barrier per stage behind the third of its four 8-row steps, the next step's fragments requested behind each
//     step's first MFMA, the DMA of the stage two ahead spread over the last step; the stream runs on across work items.
// Conv operand (implicit im2col, column k = tap * Cin + ci; a 128-column tile lies inside one tap): the source row of (row m, tap) comes
// from the geometry or the ragged row map (GemmTnDesc::rowmap); rows whose tap falls outside the sequence read zeros (out-of-range offset).
// Those per-lane offsets are the only vector arithmetic of the loop (two pieces per wave and stage, computed a stage ahead).
#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace {

struct TnpArgs {
    const float* A;  // [M, N]
    const float* B;  // [M, K] or the conv source
    float* P;        // [splits][N][K]
    int M, N, K, lda, ldb;
    int conv, T_in, T_out, stride, pad, Cin;
    const int2* rowmap;
    int tiles_n, tiles_k, splits, m_per_split, xcd_order;
};

typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) float g_zero_page_tnp[4] = {0.f, 0.f, 0.f, 0.f};

struct FragTn { f32x2 a[2][2], b[2][2]; };  // [row-pair pair][32-column block]: .x = row pair t, .y = row pair t + 1

// CONV: 0 = plain rows, 1 = conv taps by geometry, 2 = conv taps by the ragged row map
template <int CONV>
__global__ __launch_bounds__(512) void gemm_tn_f32_persist_kernel(const TnpArgs a) {
    constexpr int GBN = 256, GBK = 128, NWAVE = 8, SM = 32;  // output tile (dY columns x X columns); rows of the reduction per stage
    constexpr int A_BYTES = SM * GBN * 4, B_BYTES = SM * GBK * 4, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int APW = SM / NWAVE, WPW = SM / 2 / NWAVE;  // DMA pieces per wave and stage: 4 one-row pieces of A, 2 two-row pieces of B
    constexpr int NDMA = APW + WPW;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles = a.tiles_n * a.tiles_k;
    const int total = tiles * a.splits;
    auto decode = [&](int item, int& split, int& n0, int& k0) {
        int tile;
        if (a.xcd_order) {  // splits % 8 == 0: XCD x (blocks x, x + 8, ...) takes the splits = x mod 8 - all tiles of a split share its rows in one L2
            const int x = item & 7, j = item >> 3;
            split = x + 8 * (j / tiles);
            tile = j % tiles;
        } else {
            split = item / tiles;
            tile = item % tiles;
        }
        const int tn = tile / a.tiles_k;
        n0 = tn * GBN;
        k0 = (tile - tn * a.tiles_k) * GBK;
    };

    // ---- DMA stream state
    const unsigned a_vo = (unsigned)(lane * 16);                                  // A piece = one row's 1-KiB window
    const unsigned b_vo = (unsigned)((lane >> 5) * a.ldb * 4 + (lane & 31) * 16);  // B piece = two rows' 512-byte windows (plain rows)
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(g_zero_page_tnp, 0, 0, 0x00020000), rs_b = rs_a;
    int a_so = 0, b_so = 0;  // scalar offsets of this wave's first piece at the stream's stage
    const int a_pitch = a.lda * 4, b_pitch2 = a.ldb * 8;
    int dma_st = 0, dma_nst = 0;  // next stage of the stream within its work item, and that item's stage count (even)
    // CONV: the stream's split and tap, the per-lane offsets of the next stage's two B pieces
    int cv_mbeg = 0, cv_mend = 0, cv_tap = 0, cv_base_row = 0, cv_col = 0;
    unsigned c_eff[WPW];
    auto window_row = [&](int m, int& row0, int& bits) {  // source row of tap 0 of output row m, and which taps lie inside the sequence
        if constexpr (CONV == 2) {
            const int2 rm = a.rowmap[m];
            row0 = rm.x;
            bits = rm.y;
        } else {
            const int rr = m / a.T_out, to = m - rr * a.T_out, t0 = to * a.stride - a.pad;
            row0 = rr * a.T_in + t0;
            int b = 0;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) b |= ((unsigned)(t0 + kk) < (unsigned)a.T_in) ? (1 << kk) : 0;
            bits = b;
        }
    };
    // CONV == 1: the (sequence, output step) of each piece's row, advanced by 32 rows per stage without a division
    int g_rr[WPW], g_to[WPW];
    const int g_q32 = CONV == 1 ? SM / a.T_out : 0, g_r32 = CONV == 1 ? SM % a.T_out : 0;
    // the per-lane offsets of stream stage `stage_index` of the current stream item; FIRST: stage 0 (set-up of the item)
    auto conv_offsets = [&](int stage_index, auto firstc) {
        constexpr bool FIRST = decltype(firstc)::value;
        if constexpr (CONV == 2) {
#pragma unroll
            for (int i = 0; i < WPW; ++i) {
                const int m = cv_mbeg + stage_index * SM + 2 * (wave * WPW + i) + (lane >> 5);
                int row0, bits;
                window_row(min(m, a.M - 1), row0, bits);
                const bool ok = m < cv_mend && ((bits >> cv_tap) & 1);
                c_eff[i] = ok ? (unsigned)((row0 + cv_tap - cv_base_row) * a.Cin * 4 + cv_col + (lane & 31) * 16) : OOB;
            }
        } else if constexpr (CONV == 1) {
#pragma unroll
            for (int i = 0; i < WPW; ++i) {
                const int m = cv_mbeg + stage_index * SM + 2 * (wave * WPW + i) + (lane >> 5);
                if constexpr (FIRST) {  // once per work item: the division
                    g_rr[i] = m / a.T_out;
                    g_to[i] = m - g_rr[i] * a.T_out;
                } else {
                    g_rr[i] += g_q32;
                    g_to[i] += g_r32;
                    const bool over = g_to[i] >= a.T_out;
                    g_rr[i] += over ? 1 : 0;
                    g_to[i] -= over ? a.T_out : 0;
                }
                const int ti = g_to[i] * a.stride - a.pad + cv_tap;  // the tap's step in the source sequence
                const bool ok = m < cv_mend && (unsigned)ti < (unsigned)a.T_in;
                c_eff[i] = ok ? (unsigned)((g_rr[i] * a.T_in + ti - cv_base_row) * a.Cin * 4 + cv_col + (lane & 31) * 16) : OOB;
            }
        }
    };
    auto setup_dma = [&](int item) {
        int split, n0, k0;
        decode(item, split, n0, k0);
        const int m_begin = split * a.m_per_split, m_end = min(a.M, m_begin + a.m_per_split);
        const int rows = m_end - m_begin;
        rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A + (long long)m_begin * a.lda), 0, rows * a.lda * 4, 0x00020000);
        a_so = n0 * 4 + wave * APW * a_pitch;
        if constexpr (CONV == 0) {
            rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B + (long long)m_begin * a.ldb), 0, rows * a.ldb * 4, 0x00020000);
            b_so = k0 * 4 + wave * WPW * b_pitch2;
        } else {
            cv_mbeg = m_begin;
            cv_mend = m_end;
            cv_tap = k0 / a.Cin;
            cv_col = (k0 - cv_tap * a.Cin) * 4;
            int row0, bits;
            window_row(m_begin, row0, bits);  // m_begin < M: splits are never empty
            cv_base_row = __builtin_amdgcn_readfirstlane(row0);
            // source rows ascend with m (sequences are concatenated in order): offsets are taken from the split's first window
            rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B + (long long)cv_base_row * a.Cin), 0, 0x7fffff00, 0x00020000);
            b_so = 0;
            conv_offsets(0, std::true_type{});
        }
        dma_st = 0;
        dma_nst = ((rows + SM - 1) / SM + 1) & ~1;
    };
    auto issue_piece = [&](int stage, int q) {
        char* sbase = lds + stage * STAGE_BYTES;
        if (q < APW) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(sbase + (wave * APW + q) * 1024), 16, a_vo, a_so + q * a_pitch, 0, 0);
        } else {
            const int i = q - APW;
            const unsigned vo = CONV ? c_eff[i] : b_vo;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lptr_t)(sbase + A_BYTES + (wave * WPW + i) * 1024), 16, vo, b_so + (CONV ? 0 : i * b_pitch2), 0, 0);
        }
    };
    // past the item's last stage the stream keeps issuing rows beyond the split (zeros) into a stage nobody reads
    auto issue_advance = [&]() {
        a_so += SM * a_pitch;
        if constexpr (CONV == 0) b_so += SM * a.ldb * 4;
        ++dma_st;
        conv_offsets(dma_st, std::false_type{});
    };

    // ---- fragments: lane -> column (lane & 31) of each 32-column block, row (lane >> 5) of a row pair.  Row pairs t, t + 1 of a stage
    //      come back from one ds_read2st64_b32 (offsets in units of 256 bytes); one address register per stage, operand and column block.
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
    unsigned ra[2][2], rb[2][2];  // [stage][block]
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[st][i] = lds_base + (unsigned)(st * STAGE_BYTES + fh * (GBN * 4) + (wr * 64 + i * 32 + fr) * 4);
            rb[st][i] = lds_base + (unsigned)(st * STAGE_BYTES + A_BYTES + fh * (GBK * 4) + (wc * 64 + i * 32 + fr) * 4);
        }
#define TNP_RD2(dst, addr, o0, o1) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(o0), "n"(o1) : "memory")
    // step sp of a stage = row pairs 4 sp .. 4 sp + 3 (8 rows); A rows are 1 KiB apart (a pair = 8 units of 256 B), B rows 512 B (4 units)
    auto read_step = [&ra = ra, &rb = rb](auto stc, auto spc, FragTn& f) {
        constexpr int ST = decltype(stc)::value, SP = decltype(spc)::value;
        TNP_RD2(f.a[0][0], ra[ST][0], (4 * SP) * 8, (4 * SP + 1) * 8);
        TNP_RD2(f.a[0][1], ra[ST][1], (4 * SP) * 8, (4 * SP + 1) * 8);
        TNP_RD2(f.b[0][0], rb[ST][0], (4 * SP) * 4, (4 * SP + 1) * 4);
        TNP_RD2(f.b[0][1], rb[ST][1], (4 * SP) * 4, (4 * SP + 1) * 4);
        TNP_RD2(f.a[1][0], ra[ST][0], (4 * SP + 2) * 8, (4 * SP + 3) * 8);
        TNP_RD2(f.a[1][1], ra[ST][1], (4 * SP + 2) * 8, (4 * SP + 3) * 8);
        TNP_RD2(f.b[1][0], rb[ST][0], (4 * SP + 2) * 4, (4 * SP + 3) * 4);
        TNP_RD2(f.b[1][1], rb[ST][1], (4 * SP + 2) * 4, (4 * SP + 3) * 4);
    };
#undef TNP_RD2
    static_assert((4 * 3 + 3) * 8 <= 255, "ds_read2st64 offsets are 8 bits");
    auto land = [&](FragTn& f) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])::"memory");
    };
    f32x16 acc[2][2];
    // the 16 MFMAs of one 8-row step: row pairs in ascending order, per pair the four accumulators (gemm_tn_f32_kernel's order per element)
    auto mfma_step = [&](const FragTn& f, bool first) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (first == (e == 0 && i == 0 && j == 0))
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[e >> 1][i][e & 1], f.b[e >> 1][j][e & 1], acc[i][j], 0, 0, 0);
    };
    constexpr int NMF = 16;
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;

    int item = blockIdx.x;
    if (item >= total) return;
    setup_dma(item);
#pragma unroll
    for (int q = 0; q < NDMA; ++q) issue_piece(0, q);
    issue_advance();
#pragma unroll
    for (int q = 0; q < NDMA; ++q) issue_piece(1, q);
    issue_advance();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");  // stage 0 is the older half of what is in flight
    __builtin_amdgcn_s_barrier();
    for (; item < total; item += gridDim.x) {
        int split, n0, k0;
        decode(item, split, n0, k0);
        const int next = item + gridDim.x;
        const bool has_next = next < total;
        const int npairs = dma_nst >> 1;  // (the stream is still inside this item: its stage count)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        FragTn f0, f1;
        read_step(T0{}, T0{}, f0);
        // one stage at LDS stage ST; it issues the DMA of the stream's stage two ahead (into the same LDS stage)
        auto stage = [&](auto stc) {
            constexpr int ST = decltype(stc)::value;
            using SN = std::integral_constant<int, ST ^ 1>;
            land(f0);
            mfma_step(f0, true);
            __builtin_amdgcn_sched_barrier(0);
            read_step(stc, T1{}, f1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f0, false);
            __builtin_amdgcn_sched_barrier(0);
            land(f1);
            mfma_step(f1, true);
            __builtin_amdgcn_sched_barrier(0);
            read_step(stc, T2{}, f0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f1, false);
            __builtin_amdgcn_sched_barrier(0);
            land(f0);
            mfma_step(f0, true);
            __builtin_amdgcn_sched_barrier(0);
            read_step(stc, T3{}, f1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f0, false);
            __builtin_amdgcn_sched_barrier(0);
            // the next stage has landed (and this wave's last fragments of this one); every wave: nobody reads this LDS stage any more
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            land(f1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f1, true);
            __builtin_amdgcn_sched_barrier(0);
            read_step(SN{}, T0{}, f0);  // behind the item's last stage: the next item's first stage (requested again at its start)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NDMA; ++q) issue_piece(ST, q);
            mfma_step(f1, false);
#pragma unroll
            for (int q = 0; q < NDMA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1 - 2 * NDMA, 0);
            __builtin_amdgcn_sched_barrier(0);
            issue_advance();
        };
        for (int sp = 0; sp < npairs; ++sp) {
            if (sp == npairs - 1 && has_next) setup_dma(next);  // from the item's last-but-one stage on the stream fetches the next item
            stage(T0{});
            stage(T1{});
        }
        land(f0);  // the surplus request of the last stage

        // ---- epilogue: the raw partial sums of this (tile, split).  C/D layout of the 32x32 MFMA: column = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        float* P = a.P + (long long)split * a.N * a.K;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = k0 + wc * 64 + j * 32 + fr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    P[(long long)n * a.K + k] = acc[i][j][r];
                }
            }
    }
}

template <int CONV>
int launch_tnp_t(const TnpArgs& a, hipStream_t s) {
    constexpr size_t lds = 2 * (32 * 256 * 4 + 32 * 128 * 4);  // two stages: 96 KiB
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_f32_persist_kernel<CONV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.done(dev);
    }
    const int total = a.tiles_n * a.tiles_k * a.splits;
    hipLaunchKernelGGL((gemm_tn_f32_persist_kernel<CONV>), dim3(std::min(total, sola_cu_count())), dim3(512), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

}  // namespace

int g_gemm_tn_persist = 1;  // sola_tune "gemm_tn_persist": 0 = the 128x128 one-tile kernel for every exact-f32 weight gradient (A/B)
void sola_gemm_tn_set_persist(int v) { g_gemm_tn_persist = v; }

// The row splits of the persistent kernel for an M x (N x K) weight gradient, from the sizes alone (gemm_tn_scratch_bytes sizes the partial
// sums with it): whole rounds of one work item per CU, at least 256 rows per split.  0 = not the kernel's shape.
int gemm_tn_persist_plan(int M, int N, int K) {
    if (!g_gemm_tn_persist || N % 256 != 0 || K % 128 != 0) return 0;
    const int cus = sola_cu_count();
    const int tiles = (N / 256) * (K / 128);
    const int most = std::min(64, std::max(1, M / 256));
    int best = 0;
    double best_eff = 0.0;
    auto fill = [&](int splits) {
        const long long items = (long long)tiles * splits;
        return (double)items / (double)(((items + cus - 1) / cus) * cus);
    };
    for (int rounds = 1; rounds <= 4; ++rounds) {
        const int splits = std::min(most, rounds * cus / tiles);
        if (splits < 1) continue;
        if (fill(splits) > best_eff + 0.02) { best_eff = fill(splits); best = splits; }
    }
    if (best_eff < 0.70) return 0;  // a poor fit of the grid: the one-tile kernel's finer blocks do better
    // a multiple of 8 splits lets an XCD keep a split's rows in its own L2 - taken when it costs < 4 % of the fill
    if (best >= 8 && best % 8 != 0 && fill(best & ~7) >= fill(best) - 0.04) best &= ~7;
    return best;
}
// ... and whether this launch can take it (alignment, conv form, 32-bit offsets inside a split)
int gemm_tn_persist_splits(const GemmTnDesc& d, int max_splits) {
    int best = gemm_tn_persist_plan(d.M, d.N, d.K);
    if (best <= 0 || d.lda % 4 != 0 || (!d.conv && d.ldb % 4 != 0)) return 0;
    if (d.conv && (d.conv != 1 || d.Cin % 128 != 0 || d.K % d.Cin != 0 || d.K / d.Cin > 8)) return 0;
    if ((reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.B)) & 15) return 0;
    best = std::min(best, max_splits);
    if (best < 1) return 0;
    const long long mps = (((long long)d.M + best - 1) / best + 31) / 32 * 32;
    if (mps * std::max(d.lda, d.conv ? d.Cin * std::max(1, d.stride) : d.ldb) * 4 >= 0x7fffff00LL) return 0;
    return best;
}

// partial sums P[*splits_out][N][K] (the caller folds them: gemm_tn.hip)
int launch_gemm_tn_f32_persist(const GemmTnDesc& d, int splits, float* P, int* splits_out, hipStream_t s) {
    TnpArgs a;
    a.A = d.A; a.B = d.B; a.P = P; a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldb = d.ldb;
    a.conv = d.conv; a.T_in = d.T_in; a.T_out = d.T_out; a.stride = d.stride; a.pad = d.pad; a.Cin = d.Cin;
    a.rowmap = d.conv ? d.rowmap : nullptr;
    a.tiles_n = d.N / 256;
    a.tiles_k = d.K / 128;
    a.splits = splits;
    a.m_per_split = (int)((((long long)d.M + splits - 1) / splits + 31) / 32 * 32);
    a.splits = (d.M + a.m_per_split - 1) / a.m_per_split;  // no empty split
    *splits_out = a.splits;
    a.xcd_order = (a.splits % 8 == 0 && sola_cu_count() % 8 == 0) ? 1 : 0;
    if (!d.conv) return launch_tnp_t<0>(a, s);
    return d.rowmap ? launch_tnp_t<2>(a, s) : launch_tnp_t<1>(a, s);
}
