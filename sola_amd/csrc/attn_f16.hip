// Attention core of the 16-bit storage mode (sola_set_precision(ctx, 2)): q, k, v and the output are plain _Float16
// matrices (2 bytes per element: half the HBM traffic of the f32 / split-f16 kernels of attn.hip, which this kernel is
// otherwise a restatement of), products run on v_mfma_f32_16x16x16_f16 with f32 accumulation, softmax in f32.
//
// Same single [B, N, T', D] layout and the same (outer, inner, row) / unit-table addressing as attn.hip.  One wave owns a
// 16-query tile:
//   S^T = K Q^T : A = K rows, B = Q (registers); lane (c16, g4) ends with the scores of query c16 against keys 4*g4 + {0..3} of
//                 the tile: a softmax row lives in one 16-lane column.  Round 5: the head dimension is taken in PAIRS of 16-wide
//                 chunks - lane g4 holds the 8 halfs 32 cp + 8 g4 .. + 7 of q and of the K row (one 16-byte global load / LDS read),
//                 the first four feed the pair's first MFMA, the last four its second: a dot product does not care which k slot
//                 carries which element as long as both operands agree, and the 8-byte fragment-shaped loads (16 rows x 32 bytes
//                 per instruction) were twice the instructions on the address path;
//   O^T = V^T P^T: A = V^T (four 2-byte LDS reads down a column of the row-major tile), B = P straight from the score
//                 registers; MFMA row m of a pair's first / second product is head-dim element 32 cp + 8 (m >> 2) + (m & 3) [+ 4],
//                 so a lane ends with o[q = c16][32 cp + 8 g4 .. + 7]: one 16-byte store per pair.
// K/V tiles are 64 keys x head_dim halfs (pitch + 8 halfs: conflict-free for both read patterns on the 64-bank LDS), 34 KB
// per block at head_dim 128, so four 4-wave blocks share a CU and cover each other's memory latency; longer key sequences
// take further tiles with an online softmax.  Sequences of at most 16 steps (motion attention over T') use one wave per
// (group, head) unit with a private 16-row tile, four units per block, no block-level synchronisation.
#include "kernels.h"

namespace {

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));

struct AttnHArgs {
    const _Float16 *q, *k, *v;
    _Float16* o;
    int ldq, ldk, ldv, ldo;  // halfs
    int G, H, Sq, Sk, inner, nqb;
    long long q_outer, q_inner, q_rs;
    long long k_outer, k_inner, k_rs;
    float scale;
    int* guard;
    const int4 *q_units, *k_units;
    int qpb;  // consecutive 64-query blocks of a unit per thread block (> 1 only where the unit's keys fit one tile: K / V staged once)
    // TR instantiation (round 6: the bf16 training forward on bfloat16 q / k / v, AttnDesc::in_bf16): f32 output rows (optional), the
    // bfloat16 copy the out-projection takes as its operand (optional), the log-sum-exp of the scaled scores, dropout on the probabilities
    float* o32;
    float* lse;
    DropoutCfg drop;
};
typedef short short4v __attribute__((ext_vector_type(4)));
// one 16x16x16 product on f16 or - BF - bfloat16 operands (the 16-bit values travel as _Float16-typed bit patterns either way)
template <bool BF>
__device__ __forceinline__ f32x4 mfma_16x16x16(const half4v a, const half4v b, const f32x4 c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4v, a), __builtin_bit_cast(short4v, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
template <bool BF>
__device__ __forceinline__ _Float16 cvt_16(float x) {
    if constexpr (BF) return __builtin_bit_cast(_Float16, (__bf16)x);
    else return (_Float16)x;
}

int g_attn_f16_small = 1;
int g_attn_f16_qpb = 4;  // sola_tune "attn_f16_qpb": q-blocks per block against <= 64 keys, at most (1 = every q-block stages the unit's K / V itself; < 0: exactly -v, tests)  // sola_tune "attn_f16_small": 0 = the MFMA shape for sequences of <= 4 steps too (A/B)

struct GeoH { long long q0, k0, q_rs, k_rs; int Sq, Sk; };
__device__ __forceinline__ GeoH geo_h(const AttnHArgs& a, int grp) {
    GeoH g;
    if (a.q_units) {
        const int4 qu = a.q_units[grp], ku = a.k_units[grp];
        g.q0 = __builtin_amdgcn_readfirstlane(qu.x); g.q_rs = __builtin_amdgcn_readfirstlane(qu.y); g.Sq = __builtin_amdgcn_readfirstlane(qu.z);
        g.k0 = __builtin_amdgcn_readfirstlane(ku.x); g.k_rs = __builtin_amdgcn_readfirstlane(ku.y); g.Sk = __builtin_amdgcn_readfirstlane(ku.z);
    } else {
        g.q0 = (long long)(grp / a.inner) * a.q_outer + (long long)(grp % a.inner) * a.q_inner;
        g.k0 = (long long)(grp / a.inner) * a.k_outer + (long long)(grp % a.inner) * a.k_inner;
        g.q_rs = a.q_rs; g.k_rs = a.k_rs; g.Sq = a.Sq; g.Sk = a.Sk;
    }
    return g;
}

// WPU = wave per unit (sequences of <= 16 steps): the block's four waves serve four different (group, head) units
// (Round 5, measured and removed: two 32-key LDS stages with the next tile's rows prefetched in registers, attn_fwd_f32_simple_kernel's DB
// shape - 161 vs 151 us at the 128-key inter-object site, equal at 64 keys: at four blocks per CU the other blocks already cover a tile's load.)
// BF: bfloat16 operands (v_mfma_f32_16x16x16_bf16).  TR: the training forward - the log-sum-exp of every query row, dropout on the
// probabilities (the row sum stays undropped, as in attn_simple.hip), the output as f32 rows and / or as the 16-bit rows the out-projection reads.
template <int DH, bool WPU, bool BF = false, bool TR = false>
__global__ __launch_bounds__(256, WPU ? 2 : 4) void attn_fwd_f16_kernel(const AttnHArgs a) {
    constexpr int NC = DH / 16;        // 16-wide head-dim chunks
    constexpr bool PAIR = NC % 2 == 0; // chunk pairs with 16-byte accesses (head_dim 16: the 8-byte fragment shapes)
    constexpr int LD = DH + 8;         // tile pitch in halfs
    constexpr int H8 = DH / 8;         // 16-byte pieces per row
    constexpr int TROWS = WPU ? 16 : 64;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;

    long long unit;
    int qb = 0;
    if (WPU) {
        unit = (long long)blockIdx.x * 4 + wave;
        if (unit >= (long long)a.G * a.H) return;  // whole waves leave; this mode has no block-level sync
    } else {
        const int nqg = (a.nqb + a.qpb - 1) / a.qpb;  // q-block groups per unit
        unit = blockIdx.x / nqg;
        qb = (blockIdx.x - (int)unit * nqg) * a.qpb;
    }
    const int grp = (int)(unit / a.H), h = (int)(unit - (long long)grp * a.H);
    const GeoH g = geo_h(a, grp);
    _Float16* Ks = smem_h + (WPU ? wave * 2 * TROWS * LD : 0);
    _Float16* Vs = Ks + TROWS * LD;

    const int nq_iter = WPU ? 1 : a.qpb;
    for (int qq = 0; qq < nq_iter; ++qq) {
        const int qi = (qb + qq) * 64 + (WPU ? 0 : wave * 16) + c16;
        const bool q_ok = qi < g.Sq;
        if (!WPU && (qb + qq) * 64 >= g.Sq) return;  // ragged: this unit has fewer q-blocks than the largest one (block-uniform)
        // Q fragment: chunk pair cp, lane g4: d = 32 cp + 8 g4 .. + 7 (first four: MFMA 2 cp, last four: MFMA 2 cp + 1)
        half4v qf[NC];
        {
            const _Float16* qp = a.q + (g.q0 + (long long)qi * g.q_rs) * a.ldq + h * DH + (PAIR ? 8 : 4) * g4;
            if constexpr (!PAIR) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (q_ok) qf[c] = *reinterpret_cast<const half4v*>(qp + c * 16);
                    else qf[c] = half4v{(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
                }
            }
#pragma unroll
            for (int cp = 0; cp < NC / 2; ++cp) {
                half8v q8;
#pragma unroll
                for (int j = 0; j < 8; ++j) q8[j] = (_Float16)0;
                if (q_ok) q8 = *reinterpret_cast<const half8v*>(qp + cp * 32);
                qf[2 * cp] = half4v{q8[0], q8[1], q8[2], q8[3]};
                qf[2 * cp + 1] = half4v{q8[4], q8[5], q8[6], q8[7]};
            }
        }
        f32x4 oacc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;

        for (int kt0 = 0; kt0 < g.Sk; kt0 += TROWS) {
            const int nrows = min(TROWS, g.Sk - kt0);
            const int nrows16 = (nrows + 15) & ~15;
            if (qq == 0) {  // later q-blocks of the unit (qpb > 1: the keys fit this one tile) find K / V staged
            if (!WPU && kt0 > 0) __syncthreads();  // the previous tile has been consumed by every wave
            // stage K and V rows (16 bytes per lane), rows past the sequence zero-filled
            {
                const int nthr = WPU ? 64 : 256, t0 = WPU ? lane : tid;
                for (int idx = t0; idx < nrows16 * H8; idx += nthr) {
                    const int r = idx / H8, c8 = idx - r * H8;
                    half8v kv, vv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { kv[j] = (_Float16)0; vv[j] = (_Float16)0; }
                    if (r < nrows) {
                        const long long row = g.k0 + (long long)(kt0 + r) * g.k_rs;
                        kv = *reinterpret_cast<const half8v*>(a.k + row * a.ldk + h * DH + c8 * 8);
                        vv = *reinterpret_cast<const half8v*>(a.v + row * a.ldv + h * DH + c8 * 8);
                    }
                    *reinterpret_cast<half8v*>(&Ks[r * LD + c8 * 8]) = kv;
                    *reinterpret_cast<half8v*>(&Vs[r * LD + c8 * 8]) = vv;
                }
            }
            if (WPU) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
                __syncthreads();
            }
            }
            const int ntile = nrows16 >> 4;
            f32x4 sc[TROWS / 16];
#pragma unroll
            for (int t = 0; t < TROWS / 16; ++t) {
                if (t < ntile) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    const _Float16* kp = &Ks[(t * 16 + c16) * LD + (PAIR ? 8 : 4) * g4];
                    if constexpr (!PAIR) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const half4v kf = *reinterpret_cast<const half4v*>(kp + c * 16);
                            if (c & 1) acc1 = mfma_16x16x16<BF>(kf, qf[c], acc1);
                            else acc0 = mfma_16x16x16<BF>(kf, qf[c], acc0);
                        }
                    }
#pragma unroll
                    for (int cp = 0; cp < NC / 2; ++cp) {
                        const half8v k8 = *reinterpret_cast<const half8v*>(kp + cp * 32);
                        acc0 = mfma_16x16x16<BF>(half4v{k8[0], k8[1], k8[2], k8[3]}, qf[2 * cp], acc0);
                        acc1 = mfma_16x16x16<BF>(half4v{k8[4], k8[5], k8[6], k8[7]}, qf[2 * cp + 1], acc1);
                    }
                    const int key0 = kt0 + t * 16 + 4 * g4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[t][r] = (key0 + r < g.Sk) ? (acc0[r] + acc1[r]) * a.scale : -INFINITY;
                } else {
                    sc[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
            }
            // online softmax over this key tile (row = 16-lane column c16; key slots spread over g4 and r)
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < TROWS / 16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
            float rs = 0.f;
#pragma unroll
            for (int t = 0; t < TROWS / 16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[t][r] = __expf(sc[t][r] - m_new);
                    rs += sc[t][r];
                }
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            l_run = l_run * alpha + rs;
            m_run = m_new;
#pragma unroll
            for (int c = 0; c < NC; ++c) oacc[c] *= alpha;
            if constexpr (TR) {
                if (a.drop.enabled) {  // dropout acts on the normalised probabilities: the row sum above stays undropped (attn.hip)
                    const unsigned long long rbase = ((unsigned long long)(grp * a.H + h) * g.Sq + qi) * g.Sk;
#pragma unroll
                    for (int t = 0; t < TROWS / 16; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            sc[t][r] = dropout_keep(a.drop, rbase + kt0 + t * 16 + 4 * g4 + r) ? sc[t][r] * a.drop.scale : 0.f;
                }
            }
            // O^T += V^T P^T
#pragma unroll
            for (int t = 0; t < TROWS / 16; ++t) {
                if (t < ntile) {
                    half4v pf;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pf[r] = cvt_16<BF>(sc[t][r]);
                    // MFMA row m = c16 -> element 32 cp + 8 (m >> 2) + (m & 3) [+ 4] (PAIR), else 16 c + m
                    if constexpr (PAIR) {
                        // V^T fragment = four KEYS' values of one head-dim element per lane: gfx950's transposing LDS read hands a 16-lane group the
                        // 4 x 16 block whose sixteen 8-byte pieces its lanes address - lane i names piece (key 4 g4 + (i >> 2), elements 8 (i & 3) ..
                        // + 3 of the chunk) and receives, as lane m, the four keys' values of element 8 (m >> 2) + (m & 3): ONE read where four
                        // 2-byte reads stood (128 -> 32 LDS instructions per 64-key tile and wave; tools/micro/tr_read.hip pins the lane map)
                        const _Float16* vt = &Vs[(t * 16 + 4 * g4 + (c16 >> 2)) * LD + 8 * (c16 & 3)];
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const short4v vr = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (__attribute__((address_space(3))) short4v*)(vt + (c >> 1) * 32 + (c & 1) * 4));
                            oacc[c] = mfma_16x16x16<BF>(__builtin_bit_cast(half4v, vr), pf, oacc[c]);
                        }
                    } else {
                        const _Float16* vp = &Vs[(t * 16 + 4 * g4) * LD + c16];
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            half4v vf;
#pragma unroll
                            for (int j = 0; j < 4; ++j) vf[j] = vp[j * LD + c * 16];
                            oacc[c] = mfma_16x16x16<BF>(vf, pf, oacc[c]);
                        }
                    }
                }
            }
        }
        if constexpr (TR) {
            static_assert(!TR || PAIR, "training instantiation: head_dim a multiple of 32");
            if (q_ok) {
                const float inv = 1.f / l_run;
                const long long row = g.q0 + (long long)qi * g.q_rs;
                if (a.lse && g4 == 0) a.lse[row * a.H + h] = m_run + logf(l_run);
                const long long eo = row * a.ldo + h * DH + 8 * g4;
#pragma unroll
                for (int cp = 0; cp < NC / 2; ++cp) {
                    float v8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v8[j] = oacc[2 * cp + (j >> 2)][j & 3] * inv;
                    if (a.o32) {
                        *reinterpret_cast<float4*>(a.o32 + eo + cp * 32) = make_float4(v8[0], v8[1], v8[2], v8[3]);
                        *reinterpret_cast<float4*>(a.o32 + eo + cp * 32 + 4) = make_float4(v8[4], v8[5], v8[6], v8[7]);
                    }
                    if (a.o) {
                        half8v o8;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o8[j] = cvt_16<BF>(v8[j]);
                        *reinterpret_cast<half8v*>(a.o + eo + cp * 32) = o8;
                    }
                }
            }
            continue;
        }
        if (q_ok) {
            const float inv = 1.f / l_run;
            _Float16* op = a.o + (g.q0 + (long long)qi * g.q_rs) * a.ldo + h * DH + (PAIR ? 8 : 4) * g4;
            float m = 0.f;
            if constexpr (!PAIR) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    half4v o4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = oacc[c][j] * inv;
                        o4[j] = (_Float16)v;
                        m = fmaxf(m, fabsf(v));
                    }
                    *reinterpret_cast<half4v*>(op + c * 16) = o4;
                }
            }
#pragma unroll
            for (int cp = 0; cp < NC / 2; ++cp) {
                half8v o8;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = oacc[2 * cp + (j >> 2)][j & 3] * inv;
                    o8[j] = (_Float16)v;
                    m = fmaxf(m, fabsf(v));
                }
                if (a.ldo % 8 == 0) {
                    *reinterpret_cast<half8v*>(op + cp * 32) = o8;
                } else {  // row pitch a multiple of 4 halfs only: two 8-byte stores
                    *reinterpret_cast<half4v*>(op + cp * 32) = half4v{o8[0], o8[1], o8[2], o8[3]};
                    *reinterpret_cast<half4v*>(op + cp * 32 + 4) = half4v{o8[4], o8[5], o8[6], o8[7]};
                }
            }
            if (a.guard && !(m < 65000.f)) atomicOr(a.guard, 1);
        }
    }
}

// Sequences of at most 4 steps (motion attention over T' = 4 at the headline shape), as attn_fwd_small_kernel (attn_simple.hip): a
// (track, head) unit is 4 rows each of q, k, v in and 4 of o out - pure streaming.  The MFMA shape above pads it to a 16 x 16
// tile through a wave-private LDS slice (171 us per launch at 256 samples where the f32 small kernel moves TWICE the bytes in
// 198 us).  Here a QUARTER wave owns a unit: lane c (0..15) holds the 16-byte chunk c of every row (8 halfs), the TT x TT scores are
// per-lane partial dot products summed over the 16 lanes (DPP, no LDS), softmax and the weighted sum of the v rows in f32, one
// 16-byte store per lane and output row.
__device__ __forceinline__ float row_sum16(float v) {  // all-reduce over a 16-lane row (the first half of half_sum32, common.h)
    v = SOLA_DPP_ADD(v, 0xB1);   // quad_perm [1,0,3,2]
    v = SOLA_DPP_ADD(v, 0x4E);   // quad_perm [2,3,0,1]
    v = SOLA_DPP_ADD(v, 0x141);  // row_half_mirror
    v = SOLA_DPP_ADD(v, 0x140);  // row_mirror
    return v;
}
template <int TT>
__global__ __launch_bounds__(256) void attn_fwd_small_f16_kernel(const AttnHArgs a) {
    constexpr int DH = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, qw = lane >> 4;
    const long long unit = ((long long)blockIdx.x * 4 + wave) * 4 + qw;
    const long long n_units = (long long)a.G * a.H;
    const bool live = unit < n_units;
    const long long uu = live ? unit : 0;
    const int grp = (int)(uu / a.H), h = (int)(uu - (long long)grp * a.H);
    long long q0, k0, q_rs, k_rs;
    int Sq, Sk;
    if (a.q_units) {
        const int4 qu = a.q_units[grp], ku = a.k_units[grp];
        q0 = qu.x; q_rs = qu.y; Sq = qu.z; k0 = ku.x; k_rs = ku.y; Sk = ku.z;
    } else {
        q0 = (long long)(grp / a.inner) * a.q_outer + (long long)(grp % a.inner) * a.q_inner;
        k0 = (long long)(grp / a.inner) * a.k_outer + (long long)(grp % a.inner) * a.k_inner;
        q_rs = a.q_rs; k_rs = a.k_rs; Sq = a.Sq; Sk = a.Sk;
    }
    if (!live) { Sq = 0; Sk = 0; }
    float qv[TT][8], kv[TT][8], vv[TT][8];
    // rows past the unit's length: a CLAMPED row is loaded and the VALUE zeroed (attn_simple.hip: a select of pointers goes through
    // a scratch copy of the zero constant and a flat load)
    auto ld = [&](const _Float16* base, long long row0, long long rs, int ld_, int t, int n, float (&out)[8]) {
        const half8v v = *reinterpret_cast<const half8v*>(base + (row0 + (long long)(t < n ? t : 0) * rs) * ld_ + h * DH + 8 * c);
        const bool ok = t < n;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = ok ? (float)v[j] : 0.f;
    };
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        ld(a.q, q0, q_rs, a.ldq, t, Sq, qv[t]);
        ld(a.k, k0, k_rs, a.ldk, t, Sk, kv[t]);
        ld(a.v, k0, k_rs, a.ldv, t, Sk, vv[t]);
    }
    float sc[TT][TT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            float p = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) p += qv[i][e] * kv[j][e];
            sc[i][j] = row_sum16(p) * a.scale;
        }
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        if (i >= Sq) break;
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TT; ++j) mx = j < Sk ? fmaxf(mx, sc[i][j]) : mx;
        float den = 0.f, pj[TT];
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            pj[j] = j < Sk ? __expf(sc[i][j] - mx) : 0.f;
            den += pj[j];
        }
        const float inv = 1.f / den;
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const float w = pj[j] * inv;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += w * vv[j][e];
        }
        half8v o8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o8[e] = (_Float16)o[e];
            m = fmaxf(m, fabsf(o[e]));
        }
        *reinterpret_cast<half8v*>(a.o + (q0 + (long long)i * q_rs) * a.ldo + h * DH + 8 * c) = o8;
    }
    if (a.guard && !(m < 65000.f)) atomicOr(a.guard, 1);
}

template <int DH>
int launch_h(const AttnHArgs& a0, hipStream_t s) {
    AttnHArgs a = a0;
    constexpr int LD = DH + 8;
    const long long units = (long long)a.G * a.H;
    if (DH == 128 && a.Sq <= 4 && a.Sk <= 4 && g_attn_f16_small && a.ldo % 8 == 0) {
        const unsigned blocks = (unsigned)((units + 15) / 16);
        const int need = a.Sq > a.Sk ? a.Sq : a.Sk;
        if (need <= 1) hipLaunchKernelGGL((attn_fwd_small_f16_kernel<1>), dim3(blocks), dim3(256), 0, s, a);
        else if (need <= 2) hipLaunchKernelGGL((attn_fwd_small_f16_kernel<2>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_fwd_small_f16_kernel<4>), dim3(blocks), dim3(256), 0, s, a);
    } else if (a.Sq <= 16 && a.Sk <= 16) {
        a.nqb = 1; a.qpb = 1;
        const size_t lds = (size_t)4 * 2 * 16 * LD * sizeof(_Float16);
        hipLaunchKernelGGL((attn_fwd_f16_kernel<DH, true>), dim3((unsigned)((units + 3) / 4)), dim3(256), lds, s, a);
    } else {
        a.nqb = (a.Sq + 63) / 64;
        // many queries against at most one tile of keys (object -> language): a block walks qpb q-blocks of its unit over the staged K / V
        // as long as the grid still fills the chip several times over
        a.qpb = 1;
        if (a.Sk <= 64 && g_attn_f16_qpb < 0)  // tests: forced, whatever the grid
            a.qpb = std::min(-g_attn_f16_qpb, a.nqb);
        else if (a.Sk <= 64 && g_attn_f16_qpb > 1)
            while (a.qpb < g_attn_f16_qpb && a.qpb * 2 <= a.nqb && units * ((a.nqb + 2 * a.qpb - 1) / (2 * a.qpb)) >= 8ll * sola_cu_count()) a.qpb *= 2;
        const long long blocks = units * ((a.nqb + a.qpb - 1) / a.qpb);
        SOLA_ARG(blocks < (1ll << 31), "attention (f16): grid too large");
        const size_t lds = (size_t)2 * 64 * LD * sizeof(_Float16);
        hipLaunchKernelGGL((attn_fwd_f16_kernel<DH, false>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    }
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

// training forward on bfloat16 q / k / v (head_dim 128 / 64): one wave per unit for sequences of <= 16 steps, else 64-query blocks
template <int DH>
int launch_h_bf16_train(const AttnHArgs& a0, hipStream_t s) {
    AttnHArgs a = a0;
    constexpr int LD = DH + 8;
    const long long units = (long long)a.G * a.H;
    if (a.Sq <= 16 && a.Sk <= 16) {
        a.nqb = 1; a.qpb = 1;
        const size_t lds = (size_t)4 * 2 * 16 * LD * sizeof(_Float16);
        hipLaunchKernelGGL((attn_fwd_f16_kernel<DH, true, true, true>), dim3((unsigned)((units + 3) / 4)), dim3(256), lds, s, a);
    } else {
        a.nqb = (a.Sq + 63) / 64;
        a.qpb = 1;
        if (a.Sk <= 64 && g_attn_f16_qpb > 1)
            while (a.qpb < g_attn_f16_qpb && a.qpb * 2 <= a.nqb && units * ((a.nqb + 2 * a.qpb - 1) / (2 * a.qpb)) >= 8ll * sola_cu_count()) a.qpb *= 2;
        const long long blocks = units * ((a.nqb + a.qpb - 1) / a.qpb);
        SOLA_ARG(blocks < (1ll << 31), "attention (bf16): grid too large");
        const size_t lds = (size_t)2 * 64 * LD * sizeof(_Float16);
        hipLaunchKernelGGL((attn_fwd_f16_kernel<DH, false, true, true>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    }
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

}  // namespace

int g_attn_bf16_mfma = 1;  // sola_tune "attn_bf16_mfma": 1 = bf16 training forward on the bf16 MFMA (this file); 0 = the f32-MFMA kernel on widened values (attn_simple.hip)
void sola_attn_set_bf16_mfma(int v) { g_attn_bf16_mfma = v; }
// AttnDesc::in_bf16 launches this kernel takes: head_dim 128 / 64, 16-byte aligned rows, no shared keys, f32 and / or bf16 (o_cast_fmt 3) output
bool attention_bf16_mfma_supported(const AttnDesc& d) {
    return g_attn_bf16_mfma && d.in_bf16 && !d.o_sp16 && !d.in_sp16 && !d.k_private && (d.DH == 128 || d.DH == 64) && d.ldq % 8 == 0 && d.ldk % 8 == 0 &&
           d.ldv % 8 == 0 && d.ldo % 8 == 0 && (!d.o_cast || d.o_cast_fmt == 3) && (d.o || d.o_cast);
}
int launch_attention_bf16_train(const AttnDesc& d, hipStream_t s) {
    SOLA_ARG(d.G > 0 && d.H > 0 && d.Sq > 0 && d.Sk > 0 && d.inner > 0 && attention_bf16_mfma_supported(d), "attention (bf16): unsupported launch");
    AttnHArgs a;
    a.q = reinterpret_cast<const _Float16*>(d.q); a.k = reinterpret_cast<const _Float16*>(d.k);
    a.v = reinterpret_cast<const _Float16*>(d.v); a.o = reinterpret_cast<_Float16*>(d.o_cast);
    a.o32 = d.o; a.lse = d.lse; a.drop = d.drop;
    a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo;
    a.G = d.G; a.H = d.H; a.Sq = d.Sq; a.Sk = d.Sk; a.inner = d.inner; a.nqb = 1;
    a.q_outer = d.q_outer; a.q_inner = d.q_inner; a.q_rs = d.q_rs;
    a.k_outer = d.k_outer; a.k_inner = d.k_inner; a.k_rs = d.k_rs;
    a.scale = d.scale; a.guard = nullptr; a.qpb = 1;
    a.q_units = d.q_units; a.k_units = d.q_units ? (d.k_units ? d.k_units : d.q_units) : nullptr;
    if (d.o_cast && d.o_cast_done) *d.o_cast_done = true;
    const double elems = (double)d.G * d.H * d.DH;
    SolaProfScope prof(SOLA_PROF_ATTN, s, 4.0 * elems * d.Sq * d.Sk, elems * (2.0 * (d.Sq + 2.0 * d.Sk) + (d.o ? 4.0 : 0.0) * d.Sq + (d.o_cast ? 2.0 : 0.0) * d.Sq));
    return d.DH == 128 ? launch_h_bf16_train<128>(a, s) : launch_h_bf16_train<64>(a, s);
}

void sola_attn_set_f16_small(int v) { g_attn_f16_small = v; }
void sola_attn_set_f16_qpb(int v) { g_attn_f16_qpb = v == 0 ? 1 : v; }

// AttnDesc with q / k / v / o pointing at _Float16 matrices and ld* counting halfs
int launch_attention_f16(const AttnDesc& d, hipStream_t s) {
    SOLA_ARG(d.G > 0 && d.H > 0 && d.Sq > 0 && d.Sk > 0 && d.inner > 0, "attention (f16): bad sizes");
    SOLA_ARG(d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 4 == 0, "attention (f16): row pitches must be multiples of 8 halfs");
    SOLA_ARG(!d.lse && !d.drop.enabled, "attention (f16): inference only (no log-sum-exp output, no dropout)");
    AttnHArgs a;
    a.q = reinterpret_cast<const _Float16*>(d.q); a.k = reinterpret_cast<const _Float16*>(d.k);
    a.v = reinterpret_cast<const _Float16*>(d.v); a.o = reinterpret_cast<_Float16*>(d.o);
    a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo;
    a.G = d.G; a.H = d.H; a.Sq = d.Sq; a.Sk = d.Sk; a.inner = d.inner; a.nqb = 1;
    a.q_outer = d.q_outer; a.q_inner = d.q_inner; a.q_rs = d.q_rs;
    a.k_outer = d.k_outer; a.k_inner = d.k_inner; a.k_rs = d.k_rs;
    a.scale = d.scale; a.guard = d.guard; a.qpb = 1;
    a.o32 = nullptr; a.lse = nullptr; a.drop = DropoutCfg{};
    a.q_units = d.q_units; a.k_units = d.q_units ? (d.k_units ? d.k_units : d.q_units) : nullptr;
    const double elems = (double)d.G * d.H * d.DH;
    SolaProfScope prof(SOLA_PROF_ATTN, s, 4.0 * elems * d.Sq * d.Sk, 2.0 * elems * (2.0 * d.Sq + 2.0 * d.Sk));
    switch (d.DH) {
        case 128: return launch_h<128>(a, s);
        case 64: return launch_h<64>(a, s);
        case 32: return launch_h<32>(a, s);
        case 16: return launch_h<16>(a, s);
        default: sola_set_error("attention (f16): head_dim %d unsupported (16/32/64/128)", d.DH); return SOLA_ERR_ARG;
    }
}
