// Split-f16 NT GEMM with direct-to-LDS staging (global_load_lds_dwordx4): the staging-bound variant of gemm.hip's
// ARITH 1 kernel.  Same math (three v_mfma_f32_32x32x16_f16 per product on (hi, lo) f16 pairs, f32 accumulate); what
// changes is how a k-tile reaches LDS:
//   * every wave issues eight (or six) 1-KiB LDS-DMA pieces per k-tile: no VGPR round trip, no ds_write, and the source
//     pointers just advance 128 B per k-tile (also for the implicit-im2col conv, see below);
//   * the DMA writes LDS linearly (wave-uniform base + lane * 16 B), so tile rows are 128 contiguous bytes (8 lanes
//     per row, 8 rows per piece) and bank conflicts are removed by an XOR swizzle of the 16-byte chunks,
//     chunk' = chunk ^ ((row >> 1) & 7), applied on the SOURCE address when loading and on the fragment address when
//     reading (the same involution on both sides).  gfx950's LDS has 64 banks (256 B/clk): a ds_read_b128 is served 16
//     lanes at a time, and 16 consecutive rows reading the same logical chunk must land in 16 different 16-byte slots of
//     the 256-byte bank line - (row & 1) picks the half, the key the slot.  (A key of row & 7, right for a 32-bank LDS,
//     measured exactly half the read rate: tools/micro/lds_bw.hip, 69 vs 116 TB/s aggregate.)
//   * conv zero-padding rows read a 16-byte zero page instead (the DMA cannot predicate data); rows beyond M / N are
//     clamped to the last valid row (their products only reach outputs the epilogue never stores).
//
// Schedule: two LDS stages, ONE barrier per k-tile placed between its two 16-wide halves.  By then every wave holds both
// halves' fragments of tile kt in registers (so its stage can take the DMA of tile kt+2) and tile kt+1 has had a whole
// tile time to land, so its first fragments are fetched under the second half's MFMAs: no LDS latency is exposed behind
// a barrier and the DMA always has one full tile of MFMA work (~1.4 us) to cover its latency.
//
// Block shapes <MI, WAVES_M, WAVES_N>:
//   <2,2,2>  128x128, 4 waves of 64x64, 64 KiB LDS, two blocks per CU   - general shape
//   <4,2,4>  256x256, 8 waves of 128x64, 128 KiB, one block per CU      - 25 % fewer fragment bytes and half the DMA
//                                                                         bytes per MFMA; used when its grid fills
//                                                                         whole rounds of one block per CU
// Measured (tools/gemm_ablate.py, tools/pmc_gemm.sh, tools/pmc_clock.sh; M=16384 N=K=1024, 2.05 GHz effective clock):
// 256x256: 107 us total = 63-66 us MFMA+LDS loop alone (matrix pipe busy 68 % of cycles) + ~15 us of imperfect DMA
// overlap + ~28 us epilogue (HBM-bound: C once, residual once; with one round of blocks nothing overlaps it).  Also
// tried and dropped: a 256x128 / 3-stage shape (DMA two tiles ahead: no faster, latency was not the bound) and
// sched_group_barrier interleaving of fragment reads (+2-5 % on 64x64 wave tiles only, subsumed by this schedule).
#include "kernels.h"

namespace {

struct GldsArgs {
    GemmProblem p[3];
    int M, N, K, lda, ldr, ldc;
    int conv, T_in, T_out, stride, pad, Cin;
    int tiles_m, tiles_n, xcd_remap;
    float out_scale;
    int r_sp16, c_sp16;
    int ablate;  // measurement only (sola_tune "gemm_ablate"): 1 = no DMA after the first tiles, 4 = no epilogue
                 // (no switch around the MFMAs: control flow there makes the compiler shuttle the accumulators AGPR<->VGPR)
};

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int GBK = 32;
constexpr int ROWB = 128;  // bytes per tile row (32 elements x 4 B)

template <int MI, int WAVES_M, int WAVES_N, bool CONV>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void gemm_nt_split_glds_kernel(const GldsArgs a) {
    constexpr int GBM = MI * 32 * WAVES_M, GBN = 64 * WAVES_N;
    constexpr int STAGES = 2;
    constexpr int STAGE_BYTES = (GBM + GBN) * ROWB;
    constexpr int NWAVE = WAVES_M * WAVES_N;
    constexpr int APW = GBM / 8 / NWAVE, WPW = GBN / 8 / NWAVE;  // 8-row DMA pieces per wave per k-tile
    static_assert(APW * NWAVE * 8 == GBM && WPW * NWAVE * 8 == GBN, "tile rows must split evenly over the waves");
    static_assert(NWAVE * 64 * 64 * 4 <= STAGES * STAGE_BYTES, "epilogue staging must fit in the stage buffers");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const GemmProblem pr = a.p[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
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

    // ---- DMA coordinates: this wave owns row groups wave*APW + i (8 rows each) of the A tile and wave*WPW + i of the
    //      W tile; lane -> (row = rg*8 + lane/8, physical 16-byte chunk = lane%8).  Each piece keeps a running source
    //      pointer that advances 128 B per k-tile: also for the implicit-im2col conv, where the receptive field of an
    //      output row is (t0*Cin + k) contiguous floats of the channels-last input, so only the zero-padding test
    //      depends on k.
    const int lrow = lane >> 3, chunk = lane & 7;
    const char* a_ptr[APW];
    const char* w_ptr[WPW];
    int a_t0[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int r = (wave * APW + i) * 8 + lrow;     // tile-local A row
        const int col_bytes = (chunk ^ ((r >> 1) & 7)) * 16;  // logical 16-byte chunk that must land in this physical slot
        const int m = min(m0 + r, a.M - 1);
        if (CONV) {
            const int rr = m / a.T_out, to = m - rr * a.T_out;
            a_t0[i] = to * a.stride - a.pad;
            a_ptr[i] = reinterpret_cast<const char*>(pr.A) + ((long long)rr * a.T_in + a_t0[i]) * a.Cin * 4 + col_bytes;
        } else {
            a_t0[i] = 0;
            a_ptr[i] = reinterpret_cast<const char*>(pr.A + (long long)m * a.lda) + col_bytes;
        }
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int r = (wave * WPW + i) * 8 + lrow;  // tile-local W row
        const int n = min(n0 + r, a.N - 1);
        w_ptr[i] = reinterpret_cast<const char*>(pr.W + (long long)n * a.K) + (chunk ^ ((r >> 1) & 7)) * 16;
    }
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    int conv_kk = 0, conv_c = 0;  // CONV: tap index and channel offset of the next k-tile to issue (uniform)

    // issues k-tiles in increasing order, one per call
    auto issue = [&](int stage) {
        char* sbase = lds + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const char* src = a_ptr[i];
            if (CONV) src = (unsigned)(a_t0[i] + conv_kk) < (unsigned)a.T_in ? src : zero;  // zero padding in time
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sbase + (wave * APW + i) * 1024), 16, 0, 0);
            a_ptr[i] += GBK * 4;
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)w_ptr[i], (lptr_t)(sbase + GBM * ROWB + (wave * WPW + i) * 1024), 16, 0, 0);
            w_ptr[i] += GBK * 4;
        }
        if (CONV) {
            conv_c += GBK;
            if (conv_c == a.Cin) { conv_c = 0; ++conv_kk; }
        }
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses of this lane; the swizzle key (row >> 1) & 7 is the same for all its rows (they differ by multiples of 32)
    const int fr = lane & 31, fh = lane >> 5, key = (lane >> 1) & 7;
    const int a_frag = (wr * MI * 32 + fr) * ROWB, w_frag = GBM * ROWB + (wc * 64 + fr) * ROWB;

    struct Frags { half8 ah[MI], al[MI], bh[2], bl[2]; };
    auto load_frags = [&](const char* sbase, int s16, Frags& f) {
        const int hi_off = (((s16 * 2 + fh) * 2) ^ key) << 4;  // physical slot of the block's hi chunk
        const int lo_off = hi_off ^ 16;                           // ... and of its lo chunk (logical chunk + 1)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char* q = sbase + w_frag + j * 32 * ROWB;
            f.bh[j] = *reinterpret_cast<const half8*>(q + hi_off);
            f.bl[j] = *reinterpret_cast<const half8*>(q + lo_off);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const char* p = sbase + a_frag + i * 32 * ROWB;
            f.ah[i] = *reinterpret_cast<const half8*>(p + hi_off);
            f.al[i] = *reinterpret_cast<const half8*>(p + lo_off);
        }
    };
    auto mfmas = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
            }
    };

    const int nk = a.K / GBK;
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int stage = 0;
    Frags f0, f1;
    load_frags(lds, 0, f0);
    if (nk > 1 && !(a.ablate & 1)) issue(1);
    constexpr int NRD = 2 * MI + 4, NMF = 6 * MI;  // fragment reads and MFMAs of one half
    for (int kt = 0; kt < nk; ++kt) {
        // Each half = one MFMA batch with the NEXT half's fragment reads issued right behind its first MFMA.  The compiler
        // waits lgkmcnt(0) in front of a batch (scalar loads in the loop keep it from counting LDS returns), so reads
        // issued just before a batch - where the scheduler puts them on its own, to shorten live ranges - stall that
        // batch for the whole LDS latency of 8 waves reading at once; behind the first MFMA they have a batch to land.
        load_frags(lds + stage * STAGE_BYTES, 1, f1);
        mfmas(f0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile kt+1 has landed (issued one tile time ago)
        __syncthreads();                                   // ... and nobody reads this stage any more
        if (kt + 2 < nk && !(a.ablate & 1)) issue(stage);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(lds + (stage ^ 1) * STAGE_BYTES, 0, f0);  // past the last tile this reads stale LDS and is never used
        mfmas(f1);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        stage ^= 1;
    }

    // ---- epilogue: the accumulators (one column per lane, 16 scattered rows) go through this wave's 16 KiB of the
    //      now idle stage buffers, 64 rows at a time, and leave as whole 16-byte row pieces (16 lanes cover one 256-byte
    //      row segment) instead of 64 strided dword stores per lane.  Bias, output scale and the residual are applied on
    //      the way out.
    if (a.ablate & 4) return;
    float* tile = reinterpret_cast<float*>(lds) + wave * (64 * 64);  // [64 rows][64 cols] f32 (256-B rows: b32 writes and b128 reads are conflict-free)
    const int col_l = lane & 31, row_l = (lane >> 5) << 2;
    const int c4 = lane & 15;    // 16-byte column piece
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
#pragma unroll
    for (int h = 0; h < MI / 2; ++h) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
                    const int col = j * 32 + col_l;
                    tile[row * 64 + col] = acc[h * 2 + i][j][r];
                }
        __syncthreads();
#pragma unroll 4
        for (int pass = 0; pass < 16; ++pass) {
            const int row = pass * 4 + rsub;
            const int m = m0 + wr * MI * 32 + h * 64 + row;
            const float4 t = *reinterpret_cast<const float4*>(&tile[row * 64 + c4 * 4]);
            if (m >= a.M) continue;
            float v[4] = {t.x * a.out_scale + bv.x, t.y * a.out_scale + bv.y, t.z * a.out_scale + bv.z, t.w * a.out_scale + bv.w};
            if (pr.R) {
                if (a.r_sp16) {
                    // 4 consecutive columns sit in one 8-wide block: hi[4] and lo[4] are two aligned 8-byte loads
                    const _Float16* rb = reinterpret_cast<const _Float16*>(pr.R + (long long)m * a.ldr + (n & ~7)) + (n & 4);
                    if (n + 3 < a.N) {
                        const half4 hh = *reinterpret_cast<const half4*>(rb), ll = *reinterpret_cast<const half4*>(rb + 8);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)hh[e] + (float)ll[e];
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
            if (a.c_sp16) {
                // 4 consecutive columns of one 8-wide block: hi[4] and lo[4] leave as two aligned 8-byte stores
                _Float16* cb = reinterpret_cast<_Float16*>(pr.C + (long long)m * a.ldc + (n & ~7)) + (n & 4);
                half4 hh, ll;
#pragma unroll
                for (int e = 0; e < 4; ++e) { hh[e] = (_Float16)v[e]; ll[e] = (_Float16)(v[e] - (float)hh[e]); }
                *reinterpret_cast<half4*>(cb) = hh;
                *reinterpret_cast<half4*>(cb + 8) = ll;
            } else if (vec_ok) {
                *reinterpret_cast<float4*>(pr.C + (long long)m * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < a.N) pr.C[(long long)m * a.ldc + n + e] = v[e];
            }
        }
        if (h + 1 < MI / 2) __syncthreads();  // this wave's reads of the staging rows are done before they are rewritten
    }
}

}  // namespace

bool gemm_split_glds_supported(const GemmDesc& d) {
    if (d.arith != 1 || d.K % GBK != 0 || d.conv == 2) return false;
    if (d.conv == 1) return d.Cin % GBK == 0;
    return d.lda % 8 == 0;
}

template <int MI, int WAVES_M, int WAVES_N, bool CONV>
static int launch_glds(GldsArgs& a, int M, int N, int nprob, hipStream_t s) {
    constexpr int GBM = MI * 32 * WAVES_M, GBN = 64 * WAVES_N;
    a.tiles_m = (M + GBM - 1) / GBM;
    a.tiles_n = (N + GBN - 1) / GBN;
    a.xcd_remap = (a.tiles_m % 8 == 0) ? 1 : 0;
    constexpr size_t lds = (size_t)2 * (GBM + GBN) * ROWB;
    static bool attr_set = false;
    if (!attr_set) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_split_glds_kernel<MI, WAVES_M, WAVES_N, CONV>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_nt_split_glds_kernel<MI, WAVES_M, WAVES_N, CONV>), dim3(a.tiles_m * a.tiles_n, 1, nprob),
                       dim3(WAVES_M * WAVES_N * 64), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

template <bool CONV>
static int launch_shape(GldsArgs& a, int shape, int M, int N, int nprob, hipStream_t s) {
    if (shape == 4) return launch_glds<4, 2, 4, CONV>(a, M, N, nprob, s);
    return launch_glds<2, 2, 2, CONV>(a, M, N, nprob, s);
}

extern int g_gemm_glds;
int gemm_split_glds_shape(const GemmDesc& d);
int g_gemm_ablate = 0;
void sola_gemm_set_ablate(int v) { g_gemm_ablate = v; }

int launch_gemm_split_glds(const GemmDesc& d, hipStream_t s) {
    GldsArgs a;
    for (int i = 0; i < 3; ++i) a.p[i] = d.p[i < d.nprob ? i : 0];
    a.M = d.M; a.N = d.N; a.K = d.K; a.lda = d.lda; a.ldr = d.ldr; a.ldc = d.ldc;
    a.conv = d.conv; a.T_in = d.T_in; a.T_out = d.T_out; a.stride = d.stride; a.pad = d.pad; a.Cin = d.Cin;
    a.out_scale = d.out_scale != 0.f ? d.out_scale : 1.f;
    a.r_sp16 = d.r_sp16;
    a.c_sp16 = d.c_sp16;
    a.ablate = g_gemm_ablate;
    const int shape = gemm_split_glds_shape(d);
    return d.conv == 1 ? launch_shape<true>(a, shape, d.M, d.N, d.nprob, s) : launch_shape<false>(a, shape, d.M, d.N, d.nprob, s);
}

// g_gemm_glds: 1 = 128x128 blocks, 4 = 256x256, anything else = auto: 256x256 when its grid fills whole rounds of one
// block per CU (a partial last round of 256x256 blocks costs more than the shape gains)
int gemm_split_glds_shape(const GemmDesc& d) {
    if (g_gemm_glds == 1 || g_gemm_glds == 4) return g_gemm_glds;
    const long long t = (long long)((d.M + 255) / 256) * ((d.N + 255) / 256) * d.nprob;
    return (t >= 256 && (t % 256 == 0 || t >= 2048)) ? 4 : 1;
}
