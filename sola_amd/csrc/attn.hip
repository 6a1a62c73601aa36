// Attention core softmax(q k^T * scale) v in f32 for the three attentions of an alignment layer
// (tools/attention.py:66-72 called from module/module.py:32,41,47): inter-object (sequence = the N tracks of one
// (b,t')), motion (sequence = the T' steps of one track) and object->language (all N*T' tokens against W text tokens).
//
// The activations stay in ONE layout, [B, N, T', D] token-major; each attention only changes how a (group, row) pair
// maps to a matrix row (outer/inner/row strides), so no permuted copy is ever materialised.  Heads are column slices.
//
// HBM-bound kernel: q, k, v are read once and o written once per (group, head).  One wave (64 lanes) owns a
// 16-query tile and walks the keys 16 at a time with v_mfma_f32_16x16x4_f32 (exact f32):
//   S^T = K Q^T   : A = K rows (ds_read_b128 from LDS), B = Q rows (held in registers).
//                   Lane l then holds the scores of query (l & 15) against keys 4*(l >> 4)+{0..3}: a softmax row
//                   lives in one 16-lane column, so max/sum are 3 in-lane ops + 2 xor-shuffles (16, 32).
//   O^T = V^T P^T : A = V^T (ds_read_b32 from LDS), B = P straight from the score registers (same lane, same
//                   key slots - no cross-lane movement).  Lane l ends with o[q = l & 15][16*dt + 4*(l >> 4) + {0..3}],
//                   i.e. one float4 store per 16-wide head-dim tile.
// Two staging modes:
//   shared : a block serves one (group, head): K/V are staged once per 64-key tile (LDS sized by the real key count)
//            and, when all keys fit one tile, the block loops over several 64-query blocks against the resident K/V
//            (object->language: 256..2048 queries x 48 keys).  Online softmax across tiles handles any Sk.
//   packed : sequences of <= 16 steps (motion attention over T').  16 / pow2(T') independent (group, head) units share
//            ONE 16x16 MFMA tile as a block-diagonal problem (cross-unit scores masked to -inf), so a T'=4 launch
//            issues a quarter of the waves and MFMAs and each wave moves 4x the bytes per tile.
#include "kernels.h"

namespace {

#ifndef SOLA_ATTN_LDK_PAD
#define SOLA_ATTN_LDK_PAD 4
#endif
constexpr int LDK_PAD = SOLA_ATTN_LDK_PAD;  // 8 was the 32-bank choice; gfx950 has 64 banks

struct AttnArgs {
    const float *q, *k, *v;
    float* o;
    float* lse;  // optional [q rows][H]: log-sum-exp of the scaled scores (saved for the backward)
    int ldq, ldk, ldv, ldo;
    int G, H, Sq, Sk, inner, nqb;
    long long q_outer, q_inner, q_rs;
    long long k_outer, k_inner, k_rs;
    float scale;
    DropoutCfg drop;
    int kv_rows;      // shared: LDS rows per tile = min(64, round16(Sk))
    int qsplit;       // shared: blocks per (group, head); each handles q-blocks qs, qs + qsplit, ...
    int o_sp16;       // write o as split-f16 pairs (cast.hip) for the 3 x f16 MFMA out-projection
    int sp_log2;      // packed: log2 of the per-unit slot count SP (SP = pow2 >= max(Sq, Sk)), units per tile = 16 >> sp_log2
    int in_sp16;      // shared: q, k, v are split-f16 rows (the SPLIT kernel shape)
    int* guard;       // o_sp16: range guard word (AttnDesc::guard), null = unchecked
    const int4 *q_units, *k_units;  // ragged batches: per-group (first row, row stride, length, -) (AttnDesc::q_units)
    int tile_rows;    // shared, restaged (multi-tile) path: keys per K/V tile, 64 or 32
};

// Geometry of one group: first rows, row strides and lengths of its query and key sequences.  With unit tables the values
// are loaded per group (and made wave-uniform by hand: the compiler cannot know a table entry is the same for all lanes).
struct AttnGeo { long long q0, k0, q_rs, k_rs; int Sq, Sk; };
__device__ __forceinline__ AttnGeo attn_geo(const AttnArgs& a, int grp) {
    AttnGeo g;
    if (a.q_units) {
        const int4 qu = a.q_units[grp], ku = a.k_units[grp];
        g.q0 = qu.x; g.q_rs = qu.y; g.Sq = qu.z;
        g.k0 = ku.x; g.k_rs = ku.y; g.Sk = ku.z;
    } else {
        g.q0 = (long long)(grp / a.inner) * a.q_outer + (long long)(grp % a.inner) * a.q_inner;
        g.k0 = (long long)(grp / a.inner) * a.k_outer + (long long)(grp % a.inner) * a.k_inner;
        g.q_rs = a.q_rs; g.k_rs = a.k_rs; g.Sq = a.Sq; g.Sk = a.Sk;
    }
    return g;
}
__device__ __forceinline__ AttnGeo attn_geo_uniform(const AttnArgs& a, int grp) {  // grp is wave-uniform
    AttnGeo g = attn_geo(a, grp);
    if (a.q_units) {
        g.q0 = __builtin_amdgcn_readfirstlane((int)g.q0); g.k0 = __builtin_amdgcn_readfirstlane((int)g.k0);
        g.q_rs = __builtin_amdgcn_readfirstlane((int)g.q_rs); g.k_rs = __builtin_amdgcn_readfirstlane((int)g.k_rs);
        g.Sq = __builtin_amdgcn_readfirstlane(g.Sq); g.Sk = __builtin_amdgcn_readfirstlane(g.Sk);
    }
    return g;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

// SPLIT shape of the shared mode (inference fast path): q, k, v arrive as split-f16 rows straight from the projection
// GEMM's epilogue (every 8 values = 32 bytes [hi8 | lo8], cast.hip), so nothing is converted here; every product runs as
// lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x16_f16 with f32 accumulation (~22-bit products, as in gemm_glds.hip).  The
// exact-f32 v_mfma_f32_16x16x4_f32 has 1/16 of that rate and made the kernel co-bound by the matrix pipe (218 us of MFMA
// against 215 us of HBM at N = 128).  A lane's 4 consecutive head dims are (hi4, lo4) = two 8-byte pieces of one block.
struct HL4 { half4v hi, lo; };
__device__ __forceinline__ HL4 split4(float x, float y, float z, float w) {
    HL4 r;
    const float in[4] = {x, y, z, w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        _Float16 h1, l1;
        split_f16(in[j], h1, l1);
        r.hi[j] = h1; r.lo[j] = l1;
    }
    return r;
}
__device__ __forceinline__ f32x4 mfma3(const HL4& a, const HL4& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.hi, c, 0, 0, 0);
}

// Store one query row's output tile.  op points at o[row][h*DH]; this lane holds d = 16c + 4*g4 + {0..3} of every chunk c.
template <int NC>
__device__ __forceinline__ void store_o(float* op, int g4, const f32x4 (&oacc)[NC], float inv, int sp16, int* guard) {
    if (!sp16) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            *reinterpret_cast<float4*>(op + 4 * g4 + c * 16) =
                make_float4(oacc[c][0] * inv, oacc[c][1] * inv, oacc[c][2] * inv, oacc[c][3] * inv);
        return;
    }
    // split-f16: an 8-wide block [hi8 | lo8] is held by the lane pair (g4, g4 ^ 1); each lane writes the hi and the lo halves
    // of its OWN four values as two 8-byte stores (block offset 8 * (g4 & 1), lo 16 bytes behind) - no cross-lane traffic
    // (the lanes of a pair are 16 apart, a shuffle between them goes through the LDS crossbar)
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float v[4] = {oacc[c][0] * inv, oacc[c][1] * inv, oacc[c][2] * inv, oacc[c][3] * inv};
        half4v hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            _Float16 h1, l1;
            split_f16(v[j], h1, l1);
            hi[j] = h1; lo[j] = l1;
            m = fmaxf(m, fabsf(v[j]));
        }
        char* dst = reinterpret_cast<char*>(op + c * 16 + 8 * (g4 >> 1)) + 8 * (g4 & 1);
        *reinterpret_cast<half4v*>(dst) = hi;
        *reinterpret_cast<half4v*>(dst + 16) = lo;
    }
    if (guard && !(m < 65000.f)) atomicOr(guard, 1);  // NaN fails the comparison too
}

// NW = waves per block of the shared mode (4: 64-query blocks, K/V tiles of up to 64 rows, two blocks per CU; 8: 128-query
// blocks and a resident K/V tile of up to 128 rows for units of 65..128 keys, one block per CU).  The packed mode uses 4.
template <int DH, bool PACKED, int NW, bool SPLIT, int MINB = 2>
__global__ __launch_bounds__(NW * 64, MINB) void attn_fwd_f32_kernel(const AttnArgs a) {
    constexpr int NT = NW * 64;
    constexpr int NC = DH / 16;   // 16-wide head-dim chunks
    constexpr int LDK = DH + LDK_PAD;  // K pitch: ds_read_b128, 16 key rows per 16-lane group -> 16 distinct 16-B slots of the 64-bank line needs pitch = 4 (mod 64)
    constexpr int LDV = DH + 4;   // V pitch: ds_read_b32, the two 16-lane halves of a 32-lane group are 16 banks apart
    constexpr int F4 = DH / 4;    // float4 per row
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;

    if constexpr (PACKED) {
        // ---------------------------------------------------------------- packed: 16 >> sp_log2 units per wave tile
        float* Ks = smem + wave * 16 * (LDK + LDV);
        float* Vs = Ks + 16 * LDK;
        const int SP = 1 << a.sp_log2, U = 16 >> a.sp_log2;
        const long long units = (long long)a.G * a.H;
        const long long unit0 = ((long long)blockIdx.x * 4 + wave) * U;
        auto row_unit = [&](int r, int& grp, int& h, int& j) -> bool {  // tile row -> (group, head, step)
            const long long u = unit0 + (r >> a.sp_log2);
            j = r & (SP - 1);
            const bool ok = u < units;
            grp = ok ? (int)(u / a.H) : 0;
            h = ok ? (int)(u % a.H) : 0;
            return ok;
        };
        int grp, h, jq;
        const bool unit_ok = row_unit(c16, grp, h, jq);
        const AttnGeo gq = attn_geo(a, grp);  // this lane's query row's unit
        const bool q_ok = unit_ok && jq < gq.Sq;
        const long long qrow = gq.q0 + (long long)jq * gq.q_rs;
        float4 qf[NC];
        {
            const float* qp = a.q + qrow * a.ldq + h * DH + 4 * g4;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float4 v = q_ok ? *reinterpret_cast<const float4*>(qp + c * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
                qf[c] = make_float4(v.x * a.scale, v.y * a.scale, v.z * a.scale, v.w * a.scale);
            }
        }
        // stage this wave's 16 key rows (each row may belong to a different unit)
        for (int idx = lane; idx < 16 * F4; idx += 64) {
            const int r = idx / F4, c4 = idx - r * F4;
            int kg, kh, kj;
            const bool uok = row_unit(r, kg, kh, kj);
            const AttnGeo gk = attn_geo(a, kg);
            const bool ok = uok && kj < gk.Sk;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (ok) {
                const long long row = gk.k0 + (long long)kj * gk.k_rs;
                kv = *reinterpret_cast<const float4*>(a.k + row * a.ldk + kh * DH + c4 * 4);
                vv = *reinterpret_cast<const float4*>(a.v + row * a.ldv + kh * DH + c4 * 4);
            }
            *reinterpret_cast<float4*>(&Ks[r * LDK + c4 * 4]) = kv;
            *reinterpret_cast<float4*>(&Vs[r * LDV + c4 * 4]) = vv;
        }
        __syncthreads();
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        {
            const float* kp = &Ks[c16 * LDK + 4 * g4];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float4 kf = *reinterpret_cast<const float4*>(kp + c * 16);
                if (c & 1) {
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[c].x, a1, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[c].y, a1, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[c].z, a1, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[c].w, a1, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[c].x, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[c].y, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[c].z, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[c].w, a0, 0, 0, 0);
                }
            }
        }
        // block-diagonal mask: key slot (4*g4 + r) is visible to query row c16 only inside the same unit
        float sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kr = 4 * g4 + r;
            const bool vis = (kr >> a.sp_log2) == (c16 >> a.sp_log2) && (kr & (SP - 1)) < gq.Sk;
            sc[r] = vis ? (a0[r] + a1[r]) : -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (mx == -INFINITY) mx = 0.f;  // rows of absent units: keep exp() finite, nothing is stored for them
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = __expf(sc[r] - mx);
            rs += sc[r];
        }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        if (a.drop.enabled) {
            const unsigned long long rbase = ((unsigned long long)(grp * a.H + h) * gq.Sq + jq) * gq.Sk;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sc[r] = dropout_keep(a.drop, rbase + ((4 * g4 + r) & (SP - 1))) ? sc[r] * a.drop.scale : 0.f;
        }
        f32x4 oacc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vp = &Vs[(4 * g4 + r) * LDV + c16];
#pragma unroll
            for (int c = 0; c < NC; ++c) oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[c * 16], sc[r], oacc[c], 0, 0, 0);
        }
        if (q_ok) {
            if (a.lse && g4 == 0) a.lse[qrow * a.H + h] = mx + logf(rs);
            const float inv = 1.f / rs;
            store_o<NC>(a.o + qrow * a.ldo + h * DH, g4, oacc, inv, a.o_sp16, a.guard);
        }
    } else {
        // ---------------------------------------------------------------- shared: (group, head, q-split) units, a block
        // walks units blockIdx.x, + gridDim.x, ...  While a unit is being computed, the NEXT unit's K/V tile is already in
        // flight into registers (committed to LDS at the unit boundary) and so is its first Q fragment, so neither pipe
        // waits a full memory latency per unit (one unit per block left both the matrix pipe and HBM ~35-40 % busy).
        float* Ks = smem;                       // [kv_rows][LDK]
        float* Vs = smem + a.kv_rows * LDK;     // [kv_rows][LDV]
        struct Unit { int qs, h, grp, Sq, Sk, nqb; long long qrow0, krow0, q_rs, k_rs; };
        auto decode = [&](long long u) {
            Unit c;
            c.qs = (int)(u % a.qsplit);
            u /= a.qsplit;
            c.h = (int)(u % a.H);
            c.grp = (int)(u / a.H);
            const AttnGeo g = attn_geo_uniform(a, c.grp);
            c.qrow0 = g.q0; c.krow0 = g.k0; c.q_rs = g.q_rs; c.k_rs = g.k_rs; c.Sq = g.Sq; c.Sk = g.Sk;
            c.nqb = a.q_units ? (g.Sq + NW * 16 - 1) / (NW * 16) : a.nqb;
            return c;
        };
        const long long n_units = (long long)a.G * a.H * a.qsplit;
        const bool single_tile = a.Sk <= a.kv_rows;
        const int res_rows16 = a.q_units ? a.kv_rows : (a.Sk + 15) & ~15;  // rows of the resident tile (single_tile); zero-filled past a unit's keys
        // ONE register buffer serves both prefetches: K of the next unit is in flight from the start of a unit until the
        // unit's last QK^T is done (then it replaces Ks), V of the next unit from there until the unit's end.
        constexpr int KVR = F4 / 4;  // float4 per thread for a 64-row tile
        float4 pre[KVR];
        auto fetch = [&](const float* base, int ld, const Unit& c) {
#pragma unroll
            for (int i = 0; i < KVR; ++i) {
                const int idx = tid + NT * i;
                const int r = idx / F4, c4 = idx - r * F4;
                pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < c.Sk) pre[i] = *reinterpret_cast<const float4*>(base + (c.krow0 + (long long)r * c.k_rs) * ld + c.h * DH + c4 * 4);
            }
        };
        auto put_k = [&](int r, int c4, const float4 v) { *reinterpret_cast<float4*>(&Ks[r * LDK + c4 * 4]) = v; };
        auto put_v = [&](int r, int c4, const float4 v) { *reinterpret_cast<float4*>(&Vs[r * LDV + c4 * 4]) = v; };
        auto commit = [&](bool is_k) {
#pragma unroll
            for (int i = 0; i < KVR; ++i) {
                const int idx = tid + NT * i;
                const int r = idx / F4, c4 = idx - r * F4;
                if (r < res_rows16) {
                    if (is_k) put_k(r, c4, pre[i]);
                    else put_v(r, c4, pre[i]);
                }
            }
        };
        long long u = blockIdx.x;
        Unit cur = decode(u);
        auto stage = [&](int kt0, int nrows, int nrows16) {  // multi-tile path: straight to LDS
            for (int idx = tid; idx < nrows16 * F4; idx += NT) {
                const int r = idx / F4, c4 = idx - r * F4;
                float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
                if (r < nrows) {
                    const long long row = cur.krow0 + (long long)(kt0 + r) * cur.k_rs;
                    kv = *reinterpret_cast<const float4*>(a.k + row * a.ldk + cur.h * DH + c4 * 4);
                    vv = *reinterpret_cast<const float4*>(a.v + row * a.ldv + cur.h * DH + c4 * 4);
                }
                put_k(r, c4, kv);
                put_v(r, c4, vv);
            }
        };
        auto load_q = [&](const Unit& c, int qb, float4 (&dst)[NC]) {
            const int qi = qb * (NW * 16) + wave * 16 + c16;
            const bool ok = qb < c.nqb && qi < c.Sq;
            if constexpr (SPLIT) {
                // chunk j, lane slot g4 -> block 2j + (g4 >> 1), half (g4 & 1): hi4 and lo4 are 8 bytes each, 16 bytes apart;
                // they travel in the two halves of the float4 slot
                const char* qp = reinterpret_cast<const char*>(a.q + (c.qrow0 + (long long)qi * c.q_rs) * a.ldq + c.h * DH) +
                                 (g4 >> 1) * 32 + (g4 & 1) * 8;
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    float2 hi = make_float2(0.f, 0.f), lo = hi;
                    if (ok) {
                        hi = *reinterpret_cast<const float2*>(qp + j * 64);
                        lo = *reinterpret_cast<const float2*>(qp + j * 64 + 16);
                    }
                    dst[j] = make_float4(hi.x, hi.y, lo.x, lo.y);
                }
            } else {
                const float* qp = a.q + (c.qrow0 + (long long)qi * c.q_rs) * a.ldq + c.h * DH + 4 * g4;
#pragma unroll
                for (int j = 0; j < NC; ++j) dst[j] = ok ? *reinterpret_cast<const float4*>(qp + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto as_hl4 = [](const float2 hi, const float2 lo) {
            HL4 r;
            r.hi = __builtin_bit_cast(half4v, hi);
            r.lo = __builtin_bit_cast(half4v, lo);
            return r;
        };
        // ONE Q fragment buffer: it is only needed by the QK^T phase, so the next q-block's (or next unit's) fragment is
        // loaded into it as soon as the last key tile's scores exist, and lands under the PV phase.  (scale is applied to the
        // scores, not to q.)
        float4 qf[NC];
        load_q(cur, cur.qs, qf);
        if (single_tile) {  // first unit: K straight in, V follows through the same buffer
            fetch(a.k, a.ldk, cur);
            commit(true);
            fetch(a.v, a.ldv, cur);
        }
        for (; u < n_units; u += gridDim.x) {
        const bool has_next = u + gridDim.x < n_units;
        const Unit nxt = decode(has_next ? u + gridDim.x : u);
        if (single_tile) {
            __syncthreads();  // every wave is done with the previous unit's V
            commit(false);
            __syncthreads();  // K (committed mid-way through the previous unit) and V are visible
            if (has_next) fetch(a.k, a.ldk, nxt);
        }
        const int h = cur.h, grp = cur.grp;
        const long long qrow0 = cur.qrow0;
        for (int qb = cur.qs; qb < cur.nqb; qb += a.qsplit) {
            const int qi = qb * (NW * 16) + wave * 16 + c16;
            const bool q_ok = qi < cur.Sq;
            f32x4 oacc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            float m_run = -INFINITY, l_run = 0.f;
            for (int kt0 = 0; kt0 < cur.Sk; kt0 += a.tile_rows) {
                const int nrows = min(a.tile_rows, cur.Sk - kt0);
                const int nrows16 = (nrows + 15) & ~15;
                if (!single_tile) {
                    __syncthreads();  // the previous tile has been consumed by every wave
                    stage(kt0, nrows, nrows16);
                    __syncthreads();
                }
                const int ntile = nrows16 >> 4;
                const int kbase = single_tile ? kt0 : 0;  // a resident tile holds all keys; a restaged one starts at row 0
                // ---- scores: s[t][r] = q(c16) . k(kt0 + 16t + 4*g4 + r)
                f32x4 sc[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < ntile) {
                        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                        const float* kp = &Ks[(kbase + t * 16 + c16) * LDK + 4 * g4];
                        if constexpr (SPLIT) {
                            // A = K rows as (hi4, lo4) of head dims 16c + 4*g4 .. +3: two 8-byte reads of block 2c + (g4 >> 1)
                            const char* kb = reinterpret_cast<const char*>(&Ks[(kbase + t * 16 + c16) * LDK]) + (g4 >> 1) * 32 + (g4 & 1) * 8;
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                const HL4 kf = as_hl4(*reinterpret_cast<const float2*>(kb + c * 64), *reinterpret_cast<const float2*>(kb + c * 64 + 16));
                                const HL4 qq = as_hl4(make_float2(qf[c].x, qf[c].y), make_float2(qf[c].z, qf[c].w));
                                if (c & 1) a1 = mfma3(kf, qq, a1);
                                else a0 = mfma3(kf, qq, a0);
                            }
                        } else
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const float4 kf = *reinterpret_cast<const float4*>(kp + c * 16);
                            if (c & 1) {
                                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[c].x, a1, 0, 0, 0);
                                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[c].y, a1, 0, 0, 0);
                                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[c].z, a1, 0, 0, 0);
                                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[c].w, a1, 0, 0, 0);
                            } else {
                                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[c].x, a0, 0, 0, 0);
                                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[c].y, a0, 0, 0, 0);
                                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[c].z, a0, 0, 0, 0);
                                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[c].w, a0, 0, 0, 0);
                            }
                        }
                        const int key0 = kt0 + t * 16 + 4 * g4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[t][r] = (key0 + r < cur.Sk) ? (a0[r] + a1[r]) * a.scale : -INFINITY;
                    } else {
                        sc[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    }
                }
                if (kt0 + a.tile_rows >= cur.Sk) {  // q is dead for this q-block: fetch the next fragment into the same registers
                    if (qb + a.qsplit < cur.nqb) load_q(cur, qb + a.qsplit, qf);
                    else if (has_next) load_q(nxt, nxt.qs, qf);
                }
                // ---- online softmax over this key tile (row = 16-lane column c16; key slots spread over g4 and r)
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
                float rs = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
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
                if (a.drop.enabled) {  // dropout acts on the normalised probabilities: the row sum above stays undropped
                    const unsigned long long rbase = ((unsigned long long)(grp * a.H + h) * cur.Sq + qi) * cur.Sk;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            sc[t][r] = dropout_keep(a.drop, rbase + kt0 + t * 16 + 4 * g4 + r) ? sc[t][r] * a.drop.scale : 0.f;
                }
                if (single_tile && has_next && qb + a.qsplit >= cur.nqb && kt0 + a.tile_rows >= cur.Sk) {
                    // last QK^T of this unit is done: Ks can take the next unit's K, and its V starts to travel
                    __syncthreads();
                    commit(true);
                    fetch(a.v, a.ldv, nxt);
                }
                // ---- O^T += V^T P^T
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < ntile) {
                        if constexpr (SPLIT) {
                            // A = V^T: row d = 16c + c16, k = keys 16t + 4*g4 + {0..3}: element (key, d) of the split-f16 row is the
                            // half at block d/8, slot d%8 (hi) and +16 bytes (lo); B = P from the score registers, split here
                            const HL4 ps = split4(sc[t][0], sc[t][1], sc[t][2], sc[t][3]);
                            const char* vb = reinterpret_cast<const char*>(&Vs[(kbase + t * 16 + 4 * g4) * LDV]) + (c16 >> 3) * 32 + (c16 & 7) * 2;
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                HL4 vf;
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    vf.hi[j] = *reinterpret_cast<const _Float16*>(vb + j * (LDV * 4) + c * 64);
                                    vf.lo[j] = *reinterpret_cast<const _Float16*>(vb + j * (LDV * 4) + c * 64 + 16);
                                }
                                oacc[c] = mfma3(vf, ps, oacc[c]);
                            }
                        } else
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float* vp = &Vs[(kbase + t * 16 + 4 * g4 + r) * LDV + c16];
#pragma unroll
                            for (int c = 0; c < NC; ++c)
                                oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[c * 16], sc[t][r], oacc[c], 0, 0, 0);
                        }
                    }
                }
            }
            if (q_ok) {
                if (a.lse && g4 == 0) a.lse[(qrow0 + (long long)qi * cur.q_rs) * a.H + h] = m_run + logf(l_run);
                const float inv = 1.f / l_run;
                store_o<NC>(a.o + (qrow0 + (long long)qi * cur.q_rs) * a.ldo + h * DH, g4, oacc, inv, a.o_sp16, a.guard);
            }
        }
        cur = nxt;
        }
    }
}

int g_attn_target_blocks = 512;  // resident-K/V mode: blocks per launch (measured flat from 256 to 1024, worse above)
int g_attn_resident_blocks = 512;  // shared mode: grid size cap (2 blocks per CU x 256 CUs)
int g_attn_variant = 1;  // 1 (default): packed short sequences, high-occupancy shape for units of <= 128 queries, q-block loop over
                         // resident K/V otherwise; 0: one unit per wave / one q-block per block (baseline); 2: high-occupancy shape
                         // wherever it applies; 3: never the high-occupancy shape (A/B)

template <int DH, int NW, bool SPLIT, int MINB = 2>
static int launch_shared(const AttnArgs& a, long long blocks, size_t lds, hipStream_t s) {
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_f32_kernel<DH, false, NW, SPLIT, MINB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once.done(dev);
    }
    hipLaunchKernelGGL((attn_fwd_f32_kernel<DH, false, NW, SPLIT, MINB>), dim3((unsigned)blocks), dim3(NW * 64), lds, s, a);
    return SOLA_OK;
}

template <int DH>
int launch_dh(const AttnArgs& a0, hipStream_t s) {
    AttnArgs a = a0;
    constexpr size_t row_bytes = (size_t)((DH + LDK_PAD) + (DH + 4)) * sizeof(float);
    const bool packed = a.Sq <= 16 && a.Sk <= 16;
    SOLA_ARG(!(packed && a.in_sp16), "attention: split-f16 q/k/v are not supported for sequences of <= 16 steps (packed shape)");
    SOLA_ARG(!a.in_sp16 || DH % 16 == 0, "attention: split-f16 q/k/v need head_dim %% 16 == 0");
    if (packed) {
        static DeviceOnce once;
        int dev;
        if (once.needed(&dev)) {
            SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_f32_kernel<DH, true, 4, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64 * row_bytes)));
            once.done(dev);
        }
        int need = a.Sq > a.Sk ? a.Sq : a.Sk, lg = 0;
        while ((1 << lg) < need) ++lg;
        if (g_attn_variant == 0) lg = 4;  // baseline: one unit per wave tile
        a.sp_log2 = lg;
        const int U = 16 >> lg;
        const long long units = (long long)a.G * a.H;
        const long long tiles = (units + U - 1) / U;
        a.nqb = 1; a.kv_rows = 16; a.qsplit = 1;
        dim3 grid((unsigned)((tiles + 3) / 4));
        hipLaunchKernelGGL((attn_fwd_f32_kernel<DH, true, 4, false>), grid, dim3(256), 64 * row_bytes, s, a);
    } else {
        // units of 65..128 keys: eight waves and a resident 128-row K/V tile (135 KB, one block per CU) instead of
        // restaging two 64-row tiles for every q-block
        const bool wide = g_attn_variant != 0 && a.Sk > 64 && a.Sk <= 128 && 128 * row_bytes <= 160 * 1024;
        const int qrows = wide ? 128 : 64;
        a.nqb = (a.Sq + qrows - 1) / qrows;
        a.sp_log2 = 4;
        const int r16 = (a.Sk + 15) & ~15;
        a.kv_rows = g_attn_variant == 0 ? 64 : (r16 < qrows ? r16 : qrows);
        const long long gh = (long long)a.G * a.H;
        int qsplit = a.nqb;
        if (a.q_units) {
            qsplit = 1;  // ragged: every (group, head) is one unit that walks its own q-blocks (each group has at least one)
        } else if (g_attn_variant != 0 && a.Sk <= a.kv_rows) {  // K/V resident: loop q-blocks, keep >= ~1024 blocks in the grid
            qsplit = (int)((g_attn_target_blocks + gh - 1) / gh);
            if (qsplit < 1) qsplit = 1;
            if (qsplit > a.nqb) qsplit = a.nqb;
        }
        a.qsplit = qsplit;
        long long blocks = gh * qsplit;  // units; the grid is capped at the resident block count, blocks walk the rest
        SOLA_ARG(blocks < (1ll << 31), "attention: grid too large");
        const long long resident = wide ? g_attn_resident_blocks / 2 : g_attn_resident_blocks;
        if (g_attn_variant != 0 && blocks > resident) blocks = resident;
        a.tile_rows = 64;
        const size_t lds = (size_t)a.kv_rows * row_bytes;
        if (a.in_sp16) {
            if (wide) SOLA_TRY((launch_shared<DH, 8, true>(a, blocks, lds, s)));
            else SOLA_TRY((launch_shared<DH, 4, true>(a, blocks, lds, s)));
        } else {
            if (wide) SOLA_TRY((launch_shared<DH, 8, false>(a, blocks, lds, s)));
            else SOLA_TRY((launch_shared<DH, 4, false>(a, blocks, lds, s)));
        }
    }
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

}  // namespace

int g_attn_splitm = 0;  // sola_tune "attn_splitm": 1 = f16-MFMA triples on f32 inputs in the split precision mode (measured slower:
                        // 1.49 vs 1.39 ms of attention per 256-sample step - every wave re-converts the K/V fragments it reads)
void sola_attn_set_splitm(int v) { g_attn_splitm = v; }
void sola_attn_set_variant(int v) { g_attn_variant = v; }
void sola_attn_set_target_blocks(int v) { g_attn_target_blocks = v; }

bool attention_simple_supported(const AttnDesc& d);
int launch_attention_simple(const AttnDesc& d, hipStream_t s);
bool attention_small_supported(const AttnDesc& d);
int launch_attention_small(const AttnDesc& d, hipStream_t s);
bool attention_splitm_supported(const AttnDesc& d);
int launch_attention_splitm(const AttnDesc& d, hipStream_t s);
bool attention_reg_supported(const AttnDesc& d);
int launch_attention_reg(const AttnDesc& d, hipStream_t s);
bool attention_res_supported(const AttnDesc& d);
int launch_attention_res(const AttnDesc& d, hipStream_t s);
extern int g_attn_splitm;

bool attention_bf16_mfma_supported(const AttnDesc& d);
int launch_attention_bf16_train(const AttnDesc& d, hipStream_t s);
bool attention_spin_supported(const AttnDesc& d);
int launch_attention_spin(const AttnDesc& d, hipStream_t s);
#ifdef SOLA_EXPERIMENTS  // closed experiment (lab/attn_ring.hip): EXPERIMENTS=1 builds with sola_tune "attn_ring" only
bool attention_ring_supported(const AttnDesc& d);
int launch_attention_ring(const AttnDesc& d, hipStream_t s);
extern int g_attn_ring;
#endif

int launch_attention(const AttnDesc& d, hipStream_t s) {
    SOLA_ARG(d.G > 0 && d.H > 0 && d.Sq > 0 && d.Sk > 0 && d.inner > 0, "attention: bad sizes");
    SOLA_ARG(d.ldq % 4 == 0 && d.ldk % 4 == 0 && d.ldv % 4 == 0 && d.ldo % 4 == 0, "attention: strides must be multiples of 4");
    if (d.in_bf16) {  // bf16 q / k / v (training step's 16-bit storage)
        if (attention_bf16_mfma_supported(d)) return launch_attention_bf16_train(d, s);  // bf16 MFMA products, half the bytes per load (attn_f16.hip)
        return launch_attention_simple(d, s);                                            // f32 MFMA on the widened values (attn_simple.hip; it checks)
    }
    if (d.k_private > 0) {  // shared trailing keys: one shape implements them
        SOLA_ARG(attention_shared_keys_supported(d), "attention: shared keys (k_private %d of %d) need the few-keys shape (f32 q/k/v, head_dim 128, <= 64 keys, >= 128 queries, no unit tables)", d.k_private, d.Sk);
        return launch_attention_res(d, s);
    }
    // q / k / v written as split-f16 pairs by the projection GEMMs: the high-occupancy shape for split inputs (attn_simple.hip)
    if (g_attn_variant == 1 && attention_spin_supported(d)) return launch_attention_spin(d, s);
    // Sequences of <= 4 steps (motion attention at T <= 32): the register-only streaming shape (attn_simple.hip)
    if (g_attn_variant == 1 && attention_small_supported(d)) return launch_attention_small(d, s);
    // register-only shape (attn_reg.hip): every wave on its own 16-query tile, operands straight from global memory
    if (g_attn_variant == 1 && attention_reg_supported(d)) return launch_attention_reg(d, s);
#ifdef SOLA_EXPERIMENTS
    // sola_tune "attn_ring" 2 (A/B): the ring-staged shape also where the resident-K/V shape applies
    if (g_attn_variant == 1 && g_attn_ring == 2 && attention_res_supported(d) && attention_ring_supported(d)) return launch_attention_ring(d, s);
#endif
    // few keys, many queries (object -> language): K/V resident in LDS, 8-wave blocks streaming 16-query tiles (attn_res.hip)
    if (g_attn_variant == 1 && attention_res_supported(d)) return launch_attention_res(d, s);
    // split precision mode: f16-MFMA triples instead of the exact-f32 MFMA for every longer shape (attn_simple.hip)
    if (g_attn_variant == 1 && g_attn_splitm && attention_splitm_supported(d)) return launch_attention_splitm(d, s);
    // High-occupancy shape (attn_simple.hip) where a unit has at most two 64-query blocks and at most 128 keys - the
    // inter-object attention: measured 278 vs 308 us at N = 64, 399 vs 452 us at N = 128 (tools/attn_probe.py, B = 256 / 32).
    // With many q-blocks per unit (object -> language: 206 vs 237 us) the resident-K/V loop of this file wins and stays.
    // Ragged batches take it for every shape: its q-blocks are independent blocks, so units of very different lengths
    // balance over the chip (the resident-K/V loop walks a unit's q-blocks inside one block: 1 to 31 of them per unit in
    // the object -> language attention of a MeViS-like mix).
#ifdef SOLA_EXPERIMENTS
    // EXPERIMENTS builds with sola_tune "attn_ring" 1 only (closed experiment, lab/attn_ring.hip: bit-identical, 2-9 % slower than the
    // shipped kernel below); attention_ring_supported() is false unless that key is set
    if (g_attn_variant == 1 && attention_ring_supported(d) && ((d.Sq <= 128 && d.Sk <= 128) || d.q_units)) return launch_attention_ring(d, s);
#endif
    if (attention_simple_supported(d) && (g_attn_variant == 2 || (g_attn_variant == 1 && ((d.Sq <= 128 && d.Sk <= 128) || d.q_units))))
        return launch_attention_simple(d, s);
    AttnArgs a;
    a.q = d.q; a.k = d.k; a.v = d.v; a.o = d.o;
    a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo;
    a.G = d.G; a.H = d.H; a.Sq = d.Sq; a.Sk = d.Sk; a.inner = d.inner; a.nqb = 1;
    a.q_outer = d.q_outer; a.q_inner = d.q_inner; a.q_rs = d.q_rs;
    a.k_outer = d.k_outer; a.k_inner = d.k_inner; a.k_rs = d.k_rs;
    a.scale = d.scale;
    a.lse = d.lse;
    a.drop = d.drop;
    a.o_sp16 = d.o_sp16;
    a.in_sp16 = d.in_sp16;
    a.guard = d.o_sp16 ? d.guard : nullptr;
    a.q_units = d.q_units; a.k_units = d.q_units ? (d.k_units ? d.k_units : d.q_units) : nullptr;
    a.kv_rows = 64; a.qsplit = 1; a.sp_log2 = 4; a.tile_rows = 64;
    const double elems = (double)d.G * d.H * d.DH;
    SolaProfScope prof(SOLA_PROF_ATTN, s, 4.0 * elems * d.Sq * d.Sk, 4.0 * elems * (2.0 * d.Sq + 2.0 * d.Sk));
    switch (d.DH) {
        case 128: return launch_dh<128>(a, s);
        case 64: return launch_dh<64>(a, s);
        case 32: return launch_dh<32>(a, s);
        case 16: return launch_dh<16>(a, s);
        default: sola_set_error("attention: head_dim %d unsupported (16/32/64/128)", d.DH); return SOLA_ERR_ARG;
    }
}
