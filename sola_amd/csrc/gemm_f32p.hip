// Exact-f32 NT GEMM, persistent direct-to-LDS form (round 5): C[M,N] = A[M,K] * W[N,K]^T + bias (+ residual).
//
// Same arithmetic as gemm.hip's gemm_nt_f32_kernel - v_mfma_f32_32x32x2_f32, the k values of an output element summed in the same
// order, 128 of them per inner accumulator and the inner accumulators added to an outer one in a fixed order - so the results are
// bit-identical to that kernel (F.linear at tools/attention.py:63-65,73; F.conv1d through the implicit im2col of the channels-last
// rows, module/ws.py:14-22).  What changes is everything around the matrix pipe.  The 128x128 kernel ran at 0.75 of the f32 MFMA
// peak with the pipe idle a quarter of the time and NOT power-bound (profiles/r04_gemm_f32_pmc.txt): global -> VGPR -> LDS staging, a
// block barrier per 32-deep k-tile with the LDS latency behind it, a prologue and an epilogue per 128x128 tile.  Here:
//   * one block of eight waves per CU walks 256x128 tiles (wave tile 64x64: four 32x32 accumulators + their outer sums = 128
//     registers, two waves per SIMD);
//   * a k-tile reaches LDS by global_load_lds_dwordx4 (no VGPR round trip, no ds_write); tile rows are 128 contiguous bytes with
//     the 16-byte chunks XOR-swizzled by (row >> 1) & 7 on the SOURCE address, so the fragment reads (ds_read_b128: four k values
//     of one row per lane, lanes 0-31 the even chunk of a k-step, lanes 32-63 the odd one) are conflict-free on the 64-bank LDS;
//   * two 48-KiB stages, ONE barrier per k-tile, placed behind the third of its four 8-deep k-steps: by then every wave holds the
//     last step's fragments in registers, so the stage can take the DMA of k-tile kt+2, and k-tile kt+1 - issued a tile time
//     (~3.4 us) earlier - has landed, so its first fragments are fetched under the last step's MFMAs.  No LDS or HBM latency sits
//     behind the barrier; a wave that waits there leaves the pipe to the SIMD's other wave;
//   * the k-tiles of all of a block's tiles form one DMA stream (the next tile's first two k-tiles arrive under this tile's last
//     two), and the epilogue goes through a wave-private LDS strip (16-byte row stores, bias and residual fused) whose stores
//     drain under the next tile's MFMAs.
// Conv rows (CONV): the receptive field of an output row is contiguous in the channels-last input, so the source pointers just
// advance 128 B per k-tile; taps outside the sequence read a zero page.  Ragged batches pass the row map of GemmDesc::rowmap.
#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace {

struct F32pArgs {
    GemmProblem p[3];
    int M, N, K, lda, ldr, ldc;
    int conv, T_in, T_out, stride, pad, Cin;
    const int2* rowmap;
    int tiles_m, tiles_n, xcd_remap, nprob;
    unsigned long long* trace;  // SOLA_EXPERIMENTS, ABL & 16: per (block, wave) cycle sums (k-loops, barrier waits, epilogues, tiles)
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) float g_zero_page_f32p[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int PBK = 32;     // k values per k-tile
constexpr int PROWB = 128;  // bytes per tile row

__device__ __forceinline__ int tap_bits(int t0, int T_in) {
    int bits = 0;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) bits |= ((unsigned)(t0 + kk) < (unsigned)T_in) ? (1 << kk) : 0;
    return bits;
}

struct FragF32 { f32x4 a[2], b[2]; };
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// RMODE: 0 = no residual, 1 = f32 residual [M][ldr]
// ABL (measurement, SOLA_EXPERIMENTS builds only; results are garbage): 1 = no two-level fold, 2 = no DMA in the k-loop, 4 = no epilogue,
// 16 = cycle sums to F32pArgs::trace
//
// The k-loop carries NO vector-ALU work beyond the fold's 64 adds per 128 values of k: on gfx950 a VALU instruction does not overlap
// with the f32 MFMA (tools/micro/mfma_f32_mix.hip, profiles/r05_mfma_f32_valu.txt: every v_fma costs the matrix pipe 5-8 cycles, a
// v_pk_add_f32 12-17, at any occupancy - the f32 MFMA and the packed-f32 VALU share their peak), so
//   * the fragment addresses are eight registers computed once (one per k-step and operand; stage and row block are immediates);
//   * the DMA pieces are buffer loads: descriptor (scalar) + a per-lane offset that never changes + a scalar offset that carries the
//     piece's rows and the k position - the stream advances on the scalar ALU;
//   * the first MFMA of an accumulator after a fold takes an inline zero as its C operand (no clearing moves);
//   * the epilogue's loads and stores are buffer accesses with scalar row offsets (no 64-bit address arithmetic per row).
template <bool CONV, int RMODE, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_nt_f32_persist_kernel(const F32pArgs a) {
    constexpr int GBM = 256, GBN = 128, WAVES_N = 2, NWAVE = 8, TM = 2, TN = 2;
    constexpr int STAGE_BYTES = (GBM + GBN) * PROWB;
    constexpr int APW = GBM / 8 / NWAVE, WPW = GBN / 8 / NWAVE;  // 8-row DMA pieces per wave and k-tile: 4 + 2
    constexpr int NDMA = APW + WPW;
    constexpr int STRIP_ROWS = 16;
    constexpr int NSTORE = TM * 2 * 4;  // float4 stores per lane of a fast tile's epilogue
    constexpr unsigned OOB = 0x80000000u;  // a per-lane offset beyond every descriptor's range: the load returns zeros
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int nq = a.K / (4 * PBK);  // groups of four k-tiles = 128 values of k = one inner accumulation (gemm_nt_f32_kernel's FOLD)

    // tile order: the problems of a launch and the column tiles are the inner index, so the tiles that read one 256-row block of A
    // run back to back on one XCD (block b and its tiles b, b + grid, ... stay on XCD b % 8: the grid is a multiple of 8)
    const int tiles_row = a.tiles_n * a.nprob;
    const int total = a.tiles_m * tiles_row;
    auto decode = [&](int tile, int& z, int& m0, int& n0) {
        int rt, c;
        if (a.xcd_remap) {
            const int x = tile & 7, j = tile >> 3;
            rt = x + 8 * (j / tiles_row);
            c = j % tiles_row;
        } else {
            rt = tile / tiles_row;
            c = tile % tiles_row;
        }
        z = c / a.tiles_n;
        m0 = rt * GBM;
        n0 = (c - z * a.tiles_n) * GBN;
    };

    // ---- DMA stream: this wave owns row groups wave * APW + q (8 rows each) of the A tile and wave * WPW + i of the W tile;
    //      lane -> (row = group * 8 + lane / 8, physical 16-byte chunk = lane % 8).  The swizzle key of a piece's rows,
    //      ((group * 8 + lane / 8) >> 1) & 7, depends on the piece only through its parity: two per-lane offsets per operand.
    const int lrow = lane >> 3, chunk = lane & 7;
    unsigned a_vo[2], w_vo[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int cb = (chunk ^ (lrow >> 1) ^ (par << 2)) << 4;
        a_vo[par] = (unsigned)(lrow * a.lda * 4 + cb);  // plain rows; CONV: the row part comes from the geometry / row map (c_vo)
        w_vo[par] = (unsigned)(lrow * a.K * 4 + cb);
    }
    unsigned c_vo[APW], c_eff[APW];  // CONV: per-lane offset of the piece's window start; ... with taps outside the sequence sent out of range
    int a_t0[APW];                   // CONV: tap-validity bits of the piece's row
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(g_zero_page_f32p, 0, 0, 0x00020000), rs_w = rs_a;
    int a_so = 0, w_so = 0;          // scalar offsets of this wave's first piece at the stream's k position
    const int a_pitch8 = CONV ? 0 : a.lda * 32, w_pitch8 = a.K * 32;  // bytes between pieces (8 rows)
    int conv_kk = 0, conv_c = 0;     // CONV: tap and channel of the next k-tile to issue
    int dma_kt = 0;
    const int nk = a.K / PBK;
    auto conv_taps = [&]() {  // CONV: the per-lane offsets at tap conv_kk (a vector select per piece, when the tap changes)
        if constexpr (CONV) {
#pragma unroll
            for (int q = 0; q < APW; ++q) {
                c_eff[q] = ((a_t0[q] >> conv_kk) & 1) ? c_vo[q] : OOB;
                asm volatile("" : "+v"(c_eff[q]));  // computed HERE, once per tap (left alone the compiler re-derives the select at every k-tile's DMA)
            }
        }
    };
    auto setup_dma = [&](int tile) {
        int z, m0, n0;
        decode(tile, z, m0, n0);
        const float* A = a.p[z].A;
        const float* Wt = a.p[z].W + (long long)n0 * a.K;
        rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt), 0, min(GBN, a.N - n0) * a.K * 4, 0x00020000);
        w_so = wave * WPW * w_pitch8;
        if constexpr (CONV) {
            // window start of output row m: source row of tap 0 (may lie outside its sequence: the tap bits say which taps exist).
            // Source rows ascend with m (sequences are concatenated in order), so offsets are taken from the tile's first row.
            auto window = [&](int m, int& row0, int& bits) {
                if (a.rowmap) {
                    const int2 rm = a.rowmap[m];
                    row0 = rm.x;
                    bits = rm.y;
                } else {
                    const int rr = m / a.T_out, to = m - rr * a.T_out, t0 = to * a.stride - a.pad;
                    row0 = rr * a.T_in + t0;
                    bits = tap_bits(t0, a.T_in);
                }
            };
            int base_row, base_bits;
            window(min(m0, a.M - 1), base_row, base_bits);
            base_row = __builtin_amdgcn_readfirstlane(base_row);
            rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (long long)base_row * a.Cin), 0, 0x7fffff00, 0x00020000);
#pragma unroll
            for (int q = 0; q < APW; ++q) {
                const int r = (wave * APW + q) * 8 + lrow;
                int row0, bits;
                window(min(m0 + r, a.M - 1), row0, bits);  // rows past M: clamped (never stored)
                a_t0[q] = bits;
                c_vo[q] = (unsigned)((row0 - base_row) * a.Cin * 4 + ((chunk ^ (lrow >> 1) ^ ((q & 1) << 2)) << 4));
            }
            a_so = 0;
        } else {
            rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (long long)m0 * a.lda), 0, min(GBM, a.M - m0) * a.lda * 4, 0x00020000);
            a_so = wave * APW * a_pitch8;
        }
        conv_kk = 0;
        conv_c = 0;
        dma_kt = 0;
        conv_taps();
    };
    auto issue_piece = [&](int stage, int q) {
        char* sbase = lds + stage * STAGE_BYTES;
        if (q < APW) {
            const unsigned vo = CONV ? c_eff[q] : a_vo[q & 1];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(sbase + (wave * APW + q) * 1024), 16, vo, a_so + q * a_pitch8, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(sbase + GBM * PROWB + (wave * WPW + q - APW) * 1024), 16, w_vo[(q - APW) & 1],
                                                     w_so + (q - APW) * w_pitch8, 0, 0);
        }
    };
    // past the last k-tile of the last tile the stream re-reads that k-tile into a stage nobody reads.  Returns whether the conv tap changed.
    auto issue_advance = [&]() -> bool {
        const bool more = dma_kt + 1 < nk;
        const int adv = more ? PBK * 4 : 0;
        a_so += adv;
        w_so += adv;
        ++dma_kt;
        if constexpr (CONV) {
            conv_c += more ? PBK : 0;
            const bool wrap = conv_c == a.Cin;
            conv_c = wrap ? 0 : conv_c;
            conv_kk += wrap ? 1 : 0;
            return wrap;
        }
        return false;
    };

    // ---- fragments: lane -> tile row (lane & 31) of each 32-row block, k half (lane >> 5) of an 8-deep step; the swizzle key
    //      (row >> 1) & 7 is the same for all of a lane's rows (they differ by multiples of 32).  Step ks reads physical chunk
    //      (2 ks + fh) ^ key = byte offset xoff ^ (ks << 5): one address register per step and operand, stage and row block as immediates.
    const int fr = lane & 31, fh = lane >> 5, key = (lane >> 1) & 7;
    const int a_frag = (wr * TM * 32 + fr) * PROWB, w_frag = GBM * PROWB + (wc * TN * 32 + fr) * PROWB;
    const int xoff = (fh ^ key) << 4;
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
    unsigned ra[4], rw[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        ra[ks] = lds_base + (unsigned)(a_frag + (xoff ^ (ks << 5)));
        rw[ks] = lds_base + (unsigned)(w_frag + (xoff ^ (ks << 5)));
    }
    // The fragment reads are inline asm: in front of a compiler-visible LDS read the waitcnt pass puts vmcnt(0) when LDS-DMA pieces are
    // in flight (it cannot tell the stages apart), which would cut the DMA's tile time of cover to nothing.  land() is their wait: it
    // names the fragments as in/out operands, so every use of them is ordered behind it.
#define F32P_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
    auto read_step = [&ra = ra, &rw = rw](auto stc, auto ksc, FragF32& f) {
        constexpr int ST = decltype(stc)::value, KS = decltype(ksc)::value;
        F32P_RD(f.b[0], rw[KS], ST * STAGE_BYTES);
        F32P_RD(f.b[1], rw[KS], ST * STAGE_BYTES + 32 * PROWB);
        F32P_RD(f.a[0], ra[KS], ST * STAGE_BYTES);
        F32P_RD(f.a[1], ra[KS], ST * STAGE_BYTES + 32 * PROWB);
    };
    static_assert(TM == 2 && TN == 2 && STAGE_BYTES + 32 * PROWB < 65536, "the immediates above");
    auto land = [&](FragF32& f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.b[0]), "+v"(f.b[1])::"memory"); };
    f32x16 acc[TM][TN], outer[TM][TN];
    // the 16 MFMAs of one 8-deep k-step, in gemm_nt_f32_kernel's order per accumulator: s = 0..3, lanes 0-31 k = 8 ks + s, lanes 32-63
    // k = 8 ks + 4 + s.  `first`: the step's first MFMA alone (the next fragments are requested right behind it), else the other fifteen.
    // ZERO: the step opens an inner accumulation - its s = 0 products start from an inline zero instead of a cleared register.
    auto mfma_step = [&](const FragF32& f, bool first, auto zc) {
        constexpr bool ZERO = decltype(zc)::value;
        const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (first == (s == 0 && i == 0 && j == 0))
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][s], f.b[j][s], (ZERO && s == 0) ? z16 : acc[i][j], 0, 0, 0);
    };
    constexpr int NMF = 4 * TM * TN;
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;

    int tile = blockIdx.x;
    if (tile >= total) return;
    bool prev_fast = false;  // the previous tile of this block left through the buffer-store epilogue (its store count is known)
    setup_dma(tile);
#pragma unroll
    for (int q = 0; q < NDMA; ++q) issue_piece(0, q);
    if (issue_advance()) conv_taps();
#pragma unroll
    for (int q = 0; q < NDMA; ++q) issue_piece(1, q);
    if (issue_advance()) conv_taps();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");  // k-tile 0 is the older half of what is in flight
    __builtin_amdgcn_s_barrier();
    unsigned long long tr_loop = 0, tr_epi = 0, tr_n = 0, tr_t = 0;
    for (; tile < total; tile += gridDim.x) {
        int z, m0, n0;
        decode(tile, z, m0, n0);
        const int next = tile + gridDim.x;
        const bool has_next = next < total;
        if constexpr (ABL & 16) tr_t = clock64();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) outer[i][j][r] = 0.f;

        // k-tile 0 of this tile is in LDS, stage 0 (a tile is a whole number of k-tile groups): waited for and published by the barrier of the
        // previous tile's last k-tile, or by the prologue
        FragF32 f0, f1;
        read_step(T0{}, T0{}, f0);
        // One k-tile at position KQ of its group of four (stage KQ & 1).  It issues the DMA of the stream's k-tile two ahead.
        auto ktile = [&](auto kqc, bool relaxed_wait) {
            constexpr int KQ = decltype(kqc)::value;
            using ST = std::integral_constant<int, (KQ & 1)>;
            using SN = std::integral_constant<int, ((KQ & 1) ^ 1)>;
            using ZF = std::integral_constant<bool, KQ == 0>;
            using NZ = std::integral_constant<bool, false>;
            // ---- steps 0..2: the next step's fragments are requested right behind the step's first MFMA
            land(f0);
            mfma_step(f0, true, ZF{});
            __builtin_amdgcn_sched_barrier(0);
            read_step(ST{}, T1{}, f1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f0, false, ZF{});
            __builtin_amdgcn_sched_barrier(0);
            land(f1);
            mfma_step(f1, true, NZ{});
            __builtin_amdgcn_sched_barrier(0);
            read_step(ST{}, T2{}, f0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f1, false, NZ{});
            __builtin_amdgcn_sched_barrier(0);
            land(f0);
            mfma_step(f0, true, NZ{});
            __builtin_amdgcn_sched_barrier(0);
            read_step(ST{}, T3{}, f1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f0, false, NZ{});
            __builtin_amdgcn_sched_barrier(0);
            // ---- k-tile kt + 1 has landed (and this wave's last fragments of k-tile kt).  Behind a fast tile's epilogue the first wait names
            //      the store count: vmcnt is one in-order counter and k-tile 1's DMA was issued BEFORE the epilogue's stores, so this waits for
            //      the DMA without waiting for the stores to be acknowledged
            if (relaxed_wait) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            land(f1);
            __builtin_amdgcn_s_barrier();  // ... for every wave, and nobody reads this stage any more (step 3's fragments are in registers)
            __builtin_amdgcn_sched_barrier(0);
            // ---- step 3: the next k-tile's first fragments (the other stage) behind the first MFMA, the DMA pieces of the k-tile two ahead
            //      spread over the rest (back to back they queue in the CU's one texture-address path)
            mfma_step(f1, true, NZ{});
            __builtin_amdgcn_sched_barrier(0);
            read_step(SN{}, T0{}, f0);  // behind the tile's last k-tile: the next tile's k-tile 0 (requested again at its start)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 2)) {
#pragma unroll
                for (int q = 0; q < NDMA; ++q) issue_piece(KQ & 1, q);
            }
            mfma_step(f1, false, NZ{});
#pragma unroll
            for (int q = 0; q < ((ABL & 2) ? 0 : NDMA); ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1 - ((ABL & 2) ? 0 : 2 * NDMA), 0);
            __builtin_amdgcn_sched_barrier(0);
            if (issue_advance()) {  // (a real branch: speculated, the selects would sit in every k-tile)
                asm volatile("" ::: "memory");
                conv_taps();
            }
        };
        for (int kq = 0; kq < nq; ++kq) {
            ktile(T0{}, kq == 0 && prev_fast);
            ktile(T1{}, false);
            if (kq == nq - 1 && has_next) setup_dma(next);  // from the tile's last-but-one k-tile on the stream fetches the next tile
            ktile(T2{}, false);
            ktile(T3{}, false);
            // ---- two-level sum (gemm_nt_f32_kernel's): outer += acc after every 128 values of k, in a fixed order
            if constexpr (!(ABL & 1)) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) outer[i][j][r] += acc[i][j][r];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        land(f0);  // the surplus request of the last k-tile (the next tile's first fragments are requested again above)
        if constexpr (ABL & 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) outer[i][j] = acc[i][j];
        }
        unsigned long long te = 0;
        if constexpr (ABL & 16) { te = clock64(); tr_loop += te - tr_t; ++tr_n; }
        auto trace_end = [&]() {
            if constexpr (ABL & 16) {
                tr_epi += clock64() - te;
                if (tile + (int)gridDim.x >= total && a.trace && lane == 0) {
                    unsigned long long* rec = a.trace + ((long long)blockIdx.x * NWAVE + wave) * 4;
                    rec[0] = tr_loop; rec[1] = 0; rec[2] = tr_epi; rec[3] = tr_n;
                }
            }
        };
        if constexpr (ABL & 4) {  // keep the products alive
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(outer[i][j]));
            prev_fast = false;
            trace_end();
            continue;
        }

        // ---- epilogue: four 16-row strips per wave tile through this wave's private LDS strip (above the stages), 16-byte row pieces
        const GemmProblem pr = a.p[z];
        const float osc = pr.scale_dev ? *pr.scale_dev : 1.f;
        float* strip = reinterpret_cast<float*>(lds + 2 * STAGE_BYTES) + wave * (STRIP_ROWS * 64);
        const int col_l = lane & 31;
        const int c4 = lane & 15, rsub = lane >> 4;
        const int n = n0 + wc * 64 + c4 * 4;
        const bool vec_ok = (a.ldc & 3) == 0 && n + 3 < a.N && (RMODE != 1 || (a.ldr & 3) == 0);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pr.bias) {
            bv.x = n < a.N ? pr.bias[n] : 0.f;
            bv.y = n + 1 < a.N ? pr.bias[n + 1] : 0.f;
            bv.z = n + 2 < a.N ? pr.bias[n + 2] : 0.f;
            bv.w = n + 3 < a.N ? pr.bias[n + 3] : 0.f;
        }
        // rows of a 32x32 accumulator block held by this lane: (r & 3) + 8 * (r >> 2) + 4 * fh; r = hf * 8 .. hf * 8 + 7 are strip rows
        // (q & 3) + 8 * (q >> 2) + 4 * fh.  Bit 2 of the strip row (= fh) flips the column's bit 5 so the two half waves write different banks.
        auto to_strip = [&](int i, int hf) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int row = (q & 3) + 8 * (q >> 2) + 4 * fh;
                    const int col = (j * 32 + col_l) ^ (fh << 5);
                    strip[row * 64 + col] = outer[i][j][hf * 8 + q];
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        // Fast path: every column of the tile in range and 16-byte rows.  Loads and stores are buffer accesses - descriptor of the tile
        // (rows past M fall outside it: loads return zeros, stores are dropped), a per-lane offset that never changes, the row as a scalar offset.
        const bool fast = n0 + GBN <= a.N && (a.ldc & 3) == 0 && (RMODE != 1 || (a.ldr & 3) == 0);
        prev_fast = fast;
        if (fast) {
            const int rows = min(GBM, a.M - m0);
            const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(pr.C + (long long)m0 * a.ldc + n0, 0, rows * a.ldc * 4, 0x00020000);
            const unsigned vo_c = (unsigned)((rsub * a.ldc + wc * 64 + c4 * 4) * 4);
            // vmcnt is one in-order counter for loads and stores: a residual load issued between stores would wait for the acknowledgement of
            // every store before it, so the tile's residual values are requested first, back to back (64 registers; the fragments are dead)
            f32x4 rbuf[RMODE ? TM * 2 * 4 : 1];
            if (RMODE) {
                const __amdgpu_buffer_rsrc_t rs_r =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pr.R) + (long long)m0 * a.ldr + n0, 0, rows * a.ldr * 4, 0x00020000);
                const unsigned vo_r = (unsigned)((rsub * a.ldr + wc * 64 + c4 * 4) * 4);
#pragma unroll
                for (int sp = 0; sp < TM * 2 * 4; ++sp) {
                    const int st = sp >> 2, pass = sp & 3;
                    rbuf[sp] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, vo_r, (wr * (TM * 32) + st * 16 + pass * 4) * a.ldr * 4, 0));
                }
            }
#pragma unroll
            for (int st = 0; st < TM * 2; ++st) {
                to_strip(st >> 1, st & 1);
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int row = pass * 4 + rsub;
                    const float4 t = *reinterpret_cast<const float4*>(&strip[row * 64 + ((c4 ^ ((pass & 1) << 3)) << 2)]);
                    f32x4 v = {t.x * osc + bv.x, t.y * osc + bv.y, t.z * osc + bv.z, t.w * osc + bv.w};
                    if (RMODE) {
                        const f32x4 rv = rbuf[st * 4 + pass];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[e];
                    }
                    // (the row goes into the per-lane offset, not the scalar one: with a register in the scalar-offset field the hazard recogniser
                    // assumes a store's data registers may be rewritten at once - on gfx950 that lost parts of 16-byte stores, round 5)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_c, vo_c + (unsigned)((wr * (TM * 32) + st * 16 + pass * 4) * a.ldc * 4), 0, 0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            trace_end();
            continue;
        }
#pragma unroll
        for (int st = 0; st < TM * 2; ++st) {
            to_strip(st >> 1, st & 1);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = pass * 4 + rsub;
                const int m = m0 + wr * (TM * 32) + st * 16 + row;
                const float4 t = *reinterpret_cast<const float4*>(&strip[row * 64 + ((c4 ^ ((pass & 1) << 3)) << 2)]);
                if (m >= a.M) continue;
                float v[4] = {t.x * osc + bv.x, t.y * osc + bv.y, t.z * osc + bv.z, t.w * osc + bv.w};
                if (RMODE) {
                    if (vec_ok) {
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
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        trace_end();
    }
#undef F32P_RD
}

template <bool CONV, int RMODE, int ABL = 0>
int launch_f32p_t(const F32pArgs& a, hipStream_t s) {
    constexpr size_t lds = 2 * (256 + 128) * PROWB + 8 * 16 * 64 * 4;  // two stages + the epilogue strips: 128 KiB
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_f32_persist_kernel<CONV, RMODE, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.done(dev);
    }
    const int total = a.tiles_m * a.tiles_n * a.nprob;
    const int grid = std::min(total, sola_cu_count());
    hipLaunchKernelGGL((gemm_nt_f32_persist_kernel<CONV, RMODE, ABL>), dim3(grid), dim3(512), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

}  // namespace

#ifdef SOLA_EXPERIMENTS
int g_gemm_f32p_ablate = 0;  // sola_tune "gemm_f32p_ablate" (measurement; see ABL)
unsigned long long* g_f32p_trace = nullptr;  // [256 blocks][8 waves][4], allocated on first use
extern "C" int sola_gemm_f32p_trace_read(unsigned long long* host, int words) {
    if (!g_f32p_trace) return SOLA_ERR_ARG;
    SOLA_HIP(hipDeviceSynchronize());
    SOLA_HIP(hipMemcpy(host, g_f32p_trace, (size_t)std::min(words, 256 * 8 * 12) * 8, hipMemcpyDeviceToHost));
    return SOLA_OK;
}
#endif
int g_gemm_f32_persist = 1;  // sola_tune "gemm_f32_persist": 0 = the 128x128 one-tile-per-block kernel for every exact-f32 launch (A/B)
void sola_gemm_set_f32_persist(int v) { g_gemm_f32_persist = v; }

// Shapes the persistent kernel takes: plain rows or the conv window gather (not the transposed-conv gather), 16-byte rows, whole k-tiles,
// and a grid whose rounds of one 256x128 tile per CU are at least as full as the 128x128 kernel's rounds of two blocks per CU.
bool gemm_f32_persist_applies(const GemmDesc& d) {
    if (!g_gemm_f32_persist || d.arith != 0 || d.conv > 1 || d.K % (4 * PBK) != 0) return false;  // whole groups of four k-tiles
    if (d.conv == 1 ? (d.Cin % PBK != 0 || d.K % d.Cin != 0 || d.K / d.Cin > 8) : d.lda % 4 != 0) return false;
    for (int j = 0; j < d.nprob; ++j) {
        if ((reinterpret_cast<uintptr_t>(d.p[j].A) | reinterpret_cast<uintptr_t>(d.p[j].W) | reinterpret_cast<uintptr_t>(d.p[j].C) | reinterpret_cast<uintptr_t>(d.p[j].R)) & 15) return false;
        if ((d.p[j].R != nullptr) != (d.p[0].R != nullptr)) return false;
    }
    const int cus = sola_cu_count();
    const long long tiles = (long long)((d.M + 255) / 256) * ((d.N + 127) / 128) * d.nprob;
    if (tiles < cus) return false;
    const long long rounds = (tiles + cus - 1) / cus;
    // Rounds of one 256x128 tile per CU at ~0.92 of the pipe against the one-tile kernel's 128x128 blocks, two per CU, at ~0.75: its
    // blocks are dispatched as others finish, so a partial last round costs it about its share (+ a quarter round of imbalance), while
    // the persistent kernel pays a whole round (12288 x 1024 x 1024 = 1.5 rounds: 234 us here, 230 us there; from 2 rounds on it wins)
    const long long t128 = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.nprob;
    const double one_tile = ((double)t128 / (2.0 * cus) + (t128 % (2 * cus) ? 0.25 : 0.0)) / 0.75;
    return (double)rounds / 0.92 <= one_tile;
}

int launch_gemm_f32_persist(const GemmDesc& d, hipStream_t s) {
    F32pArgs a;
    for (int i = 0; i < 3; ++i) a.p[i] = d.p[i < d.nprob ? i : 0];
    a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldr = d.ldr; a.ldc = d.ldc;
    a.conv = d.conv; a.T_in = d.T_in; a.T_out = d.T_out; a.stride = d.stride; a.pad = d.pad; a.Cin = d.Cin;
    a.rowmap = d.conv == 1 ? d.rowmap : nullptr;
    a.tiles_m = (d.M + 255) / 256;
    a.tiles_n = (d.N + 127) / 128;
    a.nprob = d.nprob;
    a.xcd_remap = (a.tiles_m % 8 == 0 && sola_cu_count() % 8 == 0) ? 1 : 0;
    const bool res = d.p[0].R != nullptr;
    a.trace = nullptr;
#ifdef SOLA_EXPERIMENTS
    if (g_gemm_f32p_ablate && !d.conv) {
        if (!g_f32p_trace) { SOLA_HIP(hipMalloc(&g_f32p_trace, 256 * 8 * 12 * 8)); SOLA_HIP(hipMemset(g_f32p_trace, 0, 256 * 8 * 12 * 8)); }
        a.trace = g_f32p_trace;
        switch (g_gemm_f32p_ablate) {
            case 1: return res ? launch_f32p_t<false, 1, 1>(a, s) : launch_f32p_t<false, 0, 1>(a, s);
            case 2: return res ? launch_f32p_t<false, 1, 2>(a, s) : launch_f32p_t<false, 0, 2>(a, s);
            case 4: return res ? launch_f32p_t<false, 1, 4>(a, s) : launch_f32p_t<false, 0, 4>(a, s);
            case 16: return res ? launch_f32p_t<false, 1, 16>(a, s) : launch_f32p_t<false, 0, 16>(a, s);
            case 7: return res ? launch_f32p_t<false, 1, 7>(a, s) : launch_f32p_t<false, 0, 7>(a, s);
            default: break;
        }
    }
#endif
    if (d.conv) return res ? launch_f32p_t<true, 1>(a, s) : launch_f32p_t<true, 0>(a, s);
    return res ? launch_f32p_t<false, 1>(a, s) : launch_f32p_t<false, 0>(a, s);
}
