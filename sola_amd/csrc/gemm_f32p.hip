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

// RMODE: 0 = no residual, 1 = f32 residual [M][ldr]
// ABL (measurement, SOLA_EXPERIMENTS builds only; results are garbage): 1 = no two-level fold, 2 = no DMA in the k-loop, 4 = no epilogue,
// 8 = no barrier in the k-loop, 16 = cycle stamps to F32pArgs::trace
template <bool CONV, int RMODE, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_nt_f32_persist_kernel(const F32pArgs a) {
    constexpr int GBM = 256, GBN = 128, WAVES_N = 2, NWAVE = 8, TM = 2, TN = 2;
    constexpr int STAGE_BYTES = (GBM + GBN) * PROWB;
    constexpr int APW = GBM / 8 / NWAVE, WPW = GBN / 8 / NWAVE;  // 8-row DMA pieces per wave and k-tile: 4 + 2
    constexpr int NDMA = APW + WPW;
    constexpr int FOLD = 4;          // k-tiles per inner accumulation (128 values of k, as gemm_nt_f32_kernel)
    constexpr int STRIP_ROWS = 16;
    constexpr int NSTORE = TM * 2 * 4;  // float4 stores per lane of an interior tile's epilogue
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int nk = a.K / PBK;

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

    // ---- DMA stream: this wave owns row groups wave * APW + i (8 rows each) of the A tile and wave * WPW + i of the W tile;
    //      lane -> (row = group * 8 + lane / 8, physical 16-byte chunk = lane % 8)
    const int lrow = lane >> 3, chunk = lane & 7;
    const char* a_ptr0;
    const char* w_ptr0;
    int a_d[APW], w_d[WPW];
    int a_t0[APW];
    int conv_kk = 0, conv_c = 0;
    int dma_kt = 0;
    const char* zero = reinterpret_cast<const char*>(g_zero_page_f32p);
    auto setup_dma = [&](int tile) {
        int z, m0, n0;
        decode(tile, z, m0, n0);
        const float* A = a.p[z].A;
        const float* Wt = a.p[z].W;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int r = (wave * APW + i) * 8 + lrow;
            const int col_bytes = (chunk ^ ((r >> 1) & 7)) * 16;  // the logical chunk that must land in this physical slot
            const int m = min(m0 + r, a.M - 1);                   // rows past M: clamped (never stored)
            const char* p;
            if (CONV) {
                if (a.rowmap) {
                    const int2 rm = a.rowmap[m];
                    a_t0[i] = rm.y;
                    p = reinterpret_cast<const char*>(A) + (long long)rm.x * a.Cin * 4 + col_bytes;
                } else {
                    const int rr = m / a.T_out, to = m - rr * a.T_out;
                    const int t0 = to * a.stride - a.pad;
                    a_t0[i] = tap_bits(t0, a.T_in);
                    p = reinterpret_cast<const char*>(A) + ((long long)rr * a.T_in + t0) * a.Cin * 4 + col_bytes;
                }
            } else {
                a_t0[i] = 0;
                p = reinterpret_cast<const char*>(A + (long long)m * a.lda) + col_bytes;
            }
            if (i == 0) a_ptr0 = p;
            a_d[i] = (int)(p - a_ptr0);
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int r = (wave * WPW + i) * 8 + lrow;
            const int n = min(n0 + r, a.N - 1);
            const char* p = reinterpret_cast<const char*>(Wt + (long long)n * a.K) + (chunk ^ ((r >> 1) & 7)) * 16;
            if (i == 0) w_ptr0 = p;
            w_d[i] = (int)(p - w_ptr0);
        }
        conv_kk = 0;
        conv_c = 0;
        dma_kt = 0;
    };
    auto issue_piece = [&](int stage, int q) {
        char* sbase = lds + stage * STAGE_BYTES;
        if (q < APW) {
            const char* src = a_ptr0 + a_d[q];
            if (CONV) src = ((a_t0[q] >> conv_kk) & 1) ? src : zero;  // zero padding in time
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + (wave * APW + q) * 1024), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((gptr_t)(w_ptr0 + w_d[q - APW]), (lptr_t)(sbase + GBM * PROWB + (wave * WPW + q - APW) * 1024), 16, 0, 0);
        }
    };
    auto issue_advance = [&]() {  // past the last k-tile of the last tile the stream re-reads that k-tile into a stage nobody reads
        const bool more = dma_kt + 1 < nk;
        const int adv = more ? PBK * 4 : 0;
        a_ptr0 += adv;
        w_ptr0 += adv;
        if (CONV) {
            conv_c += more ? PBK : 0;
            const bool wrap = conv_c == a.Cin;
            conv_c = wrap ? 0 : conv_c;
            conv_kk += wrap ? 1 : 0;
        }
        ++dma_kt;
    };
    auto issue = [&](int stage) {
#pragma unroll
        for (int q = 0; q < NDMA; ++q) issue_piece(stage, q);
        issue_advance();
    };

    // ---- fragments: lane -> tile row (lane & 31) of each 32-row block, k half (lane >> 5) of an 8-deep step; the swizzle key
    //      (row >> 1) & 7 is the same for all of a lane's rows (they differ by multiples of 32)
    const int fr = lane & 31, fh = lane >> 5, key = (lane >> 1) & 7;
    const int a_frag = (wr * TM * 32 + fr) * PROWB, w_frag = GBM * PROWB + (wc * TN * 32 + fr) * PROWB;
    const int xoff = (fh ^ key) << 4;  // step ks reads physical chunk (2 ks + fh) ^ key = byte offset xoff ^ (ks << 5)
    // The fragment reads are inline asm: in front of a compiler-visible LDS read the waitcnt pass puts vmcnt(0) when LDS-DMA pieces are
    // in flight (it cannot tell the stages apart), which would cut the DMA's tile time of cover to nothing.  land() is their wait: it
    // names the fragments as in/out operands, so every use of them is ordered behind it.
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
#define F32P_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
    auto read_step = [&](int stage, int ks, FragF32& f) {
        const unsigned sb = lds_base + (unsigned)(stage * STAGE_BYTES) + (unsigned)(xoff ^ (ks << 5));
        const unsigned wa = sb + w_frag, aa = sb + a_frag;
        F32P_RD(f.b[0], wa, 0);
        F32P_RD(f.b[1], wa, 4096);
        F32P_RD(f.a[0], aa, 0);
        F32P_RD(f.a[1], aa, 4096);
    };
#undef F32P_RD
    static_assert(TM == 2 && TN == 2 && 32 * PROWB == 4096, "the offsets above");
    auto land = [&](FragF32& f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.b[0]), "+v"(f.b[1])::"memory"); };
    f32x16 acc[TM][TN], outer[TM][TN];
    // the 16 MFMAs of one 8-deep k-step, in gemm_nt_f32_kernel's order per accumulator: s = 0..3, lanes 0-31 k = 8 ks + s, lanes 32-63
    // k = 8 ks + 4 + s.  `first`: the step's first MFMA alone (the next fragments are requested right behind it), else the other fifteen
    auto mfma_step = [&](const FragF32& f, bool first) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (first == (s == 0 && i == 0 && j == 0)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][s], f.b[j][s], acc[i][j], 0, 0, 0);
    };
    constexpr int NMF = 4 * TM * TN;

    int tile = blockIdx.x;
    if (tile >= total) return;
    bool prev_fast = false;  // the previous tile of this block left through the interior epilogue (its store count is known)
    setup_dma(tile);
    issue(0);
    issue(1);  // nk >= 2 (checked by the launcher)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");  // k-tile 0 is the older half of what is in flight
    __builtin_amdgcn_s_barrier();
    int stage = 0;
    unsigned long long tr_loop = 0, tr_wait = 0, tr_epi = 0, tr_n = 0, tr_t = 0;
    unsigned long long ph[5] = {0, 0, 0, 0, 0};  // ABL & 128: steps 0-2, DMA wait, barrier, step 3, fold
    for (; tile < total; tile += gridDim.x) {
        int z, m0, n0;
        decode(tile, z, m0, n0);
        if constexpr (ABL & (16 | 128)) tr_t = clock64();
        const int next = tile + gridDim.x;
        const bool has_next = next < total;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = outer[i][j][r] = 0.f;

        // k-tile 0 of this tile is in LDS (waited for and published by the barrier of the previous tile's last k-tile, or by the prologue)
        FragF32 f0, f1;
        read_step(stage, 0, f0);
        for (int kt = 0; kt < nk; ++kt) {
            // k-tile kt issues the DMA of k-tile kt + 2; from kt = nk - 2 on that is the next tile's stream
            if (kt == nk - 2 && has_next) setup_dma(next);
            unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;
            if constexpr (ABL & 128) p0 = clock64();
            __builtin_amdgcn_sched_barrier(0);
            // ---- steps 0..2: the next step's fragments are requested right behind the step's first MFMA
            land(f0);
            mfma_step(f0, true);
            __builtin_amdgcn_sched_barrier(0);
            read_step(stage, 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f0, false);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ABL & 64) __builtin_amdgcn_s_setprio(1);
            land(f1);
            mfma_step(f1, true);
            __builtin_amdgcn_sched_barrier(0);
            read_step(stage, 2, f0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f1, false);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ABL & 64) __builtin_amdgcn_s_setprio(0);
            land(f0);
            mfma_step(f0, true);
            __builtin_amdgcn_sched_barrier(0);
            read_step(stage, 3, f1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f0, false);
            __builtin_amdgcn_sched_barrier(0);
            // ---- k-tile kt + 1 has landed (and this wave's last fragments of k-tile kt).  At kt == 0 behind an interior tile's epilogue that
            //      is k-tile 1, issued BEFORE the epilogue's stores: vmcnt is one in-order counter, so naming the store count waits for the DMA
            //      without waiting for the stores to be acknowledged
            if constexpr (ABL & 128) p1 = clock64();
            if (kt == 0 && prev_fast) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            land(f1);
            if constexpr (ABL & 128) p2 = clock64();
            unsigned long long tw = 0;
            if constexpr (ABL & 16) tw = clock64();
            if constexpr (!(ABL & 8))
            __builtin_amdgcn_s_barrier();  // ... for every wave, and nobody reads this stage any more (step 3's fragments are in registers)
            if constexpr (ABL & 16) tr_wait += clock64() - tw;
            if constexpr (ABL & 128) p3 = clock64();
            if constexpr (ABL & 32) __builtin_amdgcn_s_setprio(2);
            if constexpr (ABL & 64) __builtin_amdgcn_s_setprio(3);
            __builtin_amdgcn_sched_barrier(0);
            // ---- step 3: the next k-tile's first fragments (the other stage) behind the first MFMA, the DMA pieces of k-tile kt + 2 spread
            //      over the rest (back to back they queue in the CU's one texture-address path)
            mfma_step(f1, true);
            __builtin_amdgcn_sched_barrier(0);
            read_step(stage ^ 1, 0, f0);  // behind the tile's last k-tile: the next tile's k-tile 0 (requested again at its start)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 2)) issue(stage);
            mfma_step(f1, false);
#pragma unroll
            for (int q = 0; q < ((ABL & 2) ? 0 : NDMA); ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1 - ((ABL & 2) ? 0 : 2 * NDMA), 0);
            __builtin_amdgcn_sched_barrier(0);
            stage ^= 1;
            if constexpr (ABL & 128) p4 = clock64();
            // ---- two-level sum (gemm_nt_f32_kernel's): after every FOLD k-tiles and behind the last one, outer += acc in a fixed order
            if (!(ABL & 1) && ((kt & (FOLD - 1)) == FOLD - 1 || kt == nk - 1)) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            outer[i][j][r] += acc[i][j][r];
                            acc[i][j][r] = 0.f;
                        }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ABL & 128) {
                asm volatile("" : "+v"(acc[0][0]));
                const unsigned long long p5 = clock64();
                ph[0] += p1 - p0; ph[1] += p2 - p1; ph[2] += p3 - p2; ph[3] += p4 - p3; ph[4] += p5 - p4;
            }
            if constexpr (ABL & 32) __builtin_amdgcn_s_setprio(0);
            if constexpr (ABL & 64) __builtin_amdgcn_s_setprio(2);
        }
        land(f0);  // the surplus request of the last k-tile (the next tile's first fragments are requested again above)
        if constexpr (ABL & 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) outer[i][j] = acc[i][j];
        }
        unsigned long long te = 0;
        if constexpr (ABL & (16 | 128)) { te = clock64(); tr_loop += te - tr_t; ++tr_n; }
        auto trace_end = [&]() {
            if constexpr (ABL & (16 | 128)) {
                tr_epi += clock64() - te;
                if (tile + (int)gridDim.x >= total && a.trace && lane == 0) {
                    unsigned long long* rec = a.trace + ((long long)blockIdx.x * NWAVE + wave) * 4;
                    rec[0] = tr_loop; rec[1] = tr_wait; rec[2] = tr_epi; rec[3] = tr_n;
                    if constexpr (ABL & 128) {
                        unsigned long long* r2 = a.trace + 256 * 8 * 4 + ((long long)blockIdx.x * NWAVE + wave) * 8;
                        for (int e = 0; e < 5; ++e) r2[e] = ph[e];
                    }
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

        // ---- epilogue: four 16-row strips per wave tile through this wave's private LDS strip (above the stages), 16-byte row stores
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
        const bool interior = m0 + GBM <= a.M && n0 + GBN <= a.N && (a.ldc & 3) == 0 && (RMODE != 1 || (a.ldr & 3) == 0);
        prev_fast = interior;
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
        if (interior) {
            // vmcnt is one in-order counter for loads and stores: a residual load issued between stores would wait for the acknowledgement of
            // every store before it, so the tile's residual values are requested first, back to back (64 registers; the fragments are dead)
            f32x4 rbuf[RMODE ? TM * 2 * 4 : 1];
            if (RMODE) {
#pragma unroll
                for (int sp = 0; sp < TM * 2 * 4; ++sp) {
                    const int st = sp >> 2, pass = sp & 3;
                    const int m = m0 + wr * (TM * 32) + st * 16 + pass * 4 + rsub;
                    rbuf[sp] = *reinterpret_cast<const f32x4*>(pr.R + (long long)m * a.ldr + n);
                }
            }
#pragma unroll
            for (int st = 0; st < TM * 2; ++st) {
                to_strip(st >> 1, st & 1);
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int row = pass * 4 + rsub;
                    const int m = m0 + wr * (TM * 32) + st * 16 + row;
                    const float4 t = *reinterpret_cast<const float4*>(&strip[row * 64 + ((c4 ^ ((pass & 1) << 3)) << 2)]);
                    float v[4] = {t.x * osc + bv.x, t.y * osc + bv.y, t.z * osc + bv.z, t.w * osc + bv.w};
                    if (RMODE) {
                        const f32x4 rv = rbuf[st * 4 + pass];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[e];
                    }
                    *reinterpret_cast<float4*>(pr.C + (long long)m * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
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
    if (!g_gemm_f32_persist || d.arith != 0 || d.conv > 1 || d.K % PBK != 0 || d.K / PBK < 2) return false;
    if (d.conv == 1 ? (d.Cin % PBK != 0 || d.K % d.Cin != 0 || d.K / d.Cin > 8) : d.lda % 4 != 0) return false;
    for (int j = 0; j < d.nprob; ++j) {
        if ((reinterpret_cast<uintptr_t>(d.p[j].A) | reinterpret_cast<uintptr_t>(d.p[j].W) | reinterpret_cast<uintptr_t>(d.p[j].C) | reinterpret_cast<uintptr_t>(d.p[j].R)) & 15) return false;
        if ((d.p[j].R != nullptr) != (d.p[0].R != nullptr)) return false;
    }
    const int cus = sola_cu_count();
    const long long tiles = (long long)((d.M + 255) / 256) * ((d.N + 127) / 128) * d.nprob;
    if (tiles < cus) return false;
    const long long rounds = (tiles + cus - 1) / cus;
    // time estimates in units of one 128x128 tile on a quarter of a CU... persistent: rounds of 256x128 tiles at ~0.92 of the pipe;
    // one-tile kernel: rounds of 2 * cus 128x128 tiles at ~0.75
    const long long t128 = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.nprob;
    const long long rounds128 = (t128 + 2 * cus - 1) / (2 * cus);
    return (double)rounds * 2.0 / 0.92 <= (double)rounds128 * 2.0 / 0.75;
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
            case 8: return res ? launch_f32p_t<false, 1, 8>(a, s) : launch_f32p_t<false, 0, 8>(a, s);
            case 32: return res ? launch_f32p_t<false, 1, 32>(a, s) : launch_f32p_t<false, 0, 32>(a, s);
            case 64: return res ? launch_f32p_t<false, 1, 64>(a, s) : launch_f32p_t<false, 0, 64>(a, s);
            case 48: return res ? launch_f32p_t<false, 1, 48>(a, s) : launch_f32p_t<false, 0, 48>(a, s);
            case 36: return res ? launch_f32p_t<false, 1, 36>(a, s) : launch_f32p_t<false, 0, 36>(a, s);
            case 128: return res ? launch_f32p_t<false, 1, 128>(a, s) : launch_f32p_t<false, 0, 128>(a, s);
            case 192: return res ? launch_f32p_t<false, 1, 192>(a, s) : launch_f32p_t<false, 0, 192>(a, s);
            case 160: return res ? launch_f32p_t<false, 1, 160>(a, s) : launch_f32p_t<false, 0, 160>(a, s);
            case 16: return res ? launch_f32p_t<false, 1, 16>(a, s) : launch_f32p_t<false, 0, 16>(a, s);
            case 7: return res ? launch_f32p_t<false, 1, 7>(a, s) : launch_f32p_t<false, 0, 7>(a, s);
            case 15: return res ? launch_f32p_t<false, 1, 15>(a, s) : launch_f32p_t<false, 0, 15>(a, s);
            default: break;
        }
    }
#endif
    if (d.conv) return res ? launch_f32p_t<true, 1>(a, s) : launch_f32p_t<true, 0>(a, s);
    return res ? launch_f32p_t<false, 1>(a, s) : launch_f32p_t<false, 0>(a, s);
}
