// High-occupancy shape of the exact-f32 attention core (inference; sola_tune "attn_variant" 2).
//
// attn.hip's shared mode keeps a whole 64-key K/V tile pair in LDS (68 KB) and 256 VGPRs per lane for its register
// prefetch of the next unit: two 4-wave blocks per CU, and a unit's phases (stage, QK^T, softmax, PV, store) largely
// serialise - it reaches 3.5 TB/s where a plain copy with the same access pattern reaches 5.3-5.9 (tools/micro/strided_bw).
// This shape trades the hand-rolled pipeline for occupancy: K/V tiles of TK = 32 keys (34 KB), no cross-unit state, ~100
// VGPRs, so four to five blocks share a CU and cover each other's latency; longer key sequences take further tiles with
// the online softmax.  Same arithmetic (v_mfma_f32_16x16x4_f32, S^T = K Q^T / O^T = V^T P^T register layout), same
// addressing (strided groups or unit tables), f32 or split-f16 output.
#include "kernels.h"

namespace {

typedef _Float16 half4v __attribute__((ext_vector_type(4)));

struct AttnSArgs {
    const float *q, *k, *v;
    float* o;
    int ldq, ldk, ldv, ldo;
    int G, H, Sq, Sk, inner, nqb;
    long long q_outer, q_inner, q_rs;
    long long k_outer, k_inner, k_rs;
    float scale;
    int o_sp16;
    int in_bf16;  // q / k / v are bfloat16 rows (IN16 instantiation)
    int* guard;
    const int4 *q_units, *k_units;
    int xcd_remap;
    float* lse;       // TR instantiation: optional [q rows][H] log-sum-exp of the scaled scores (saved for the backward)
    DropoutCfg drop;  // TR instantiation: dropout on the probabilities (tools/attention.py:71), same counter-based mask as attn.hip
    void* o_cast;     // TR instantiation: AttnDesc::o_cast / o_side / o_cast_fmt
    void* o_side;
    int o_cast_fmt;
    int o_skip_f32;   // TR + IN16: the f32 output is not written (the bf16 o_cast is the only copy: AttnDesc::in_bf16 with o == nullptr)
};

// four / eight bfloat16 values of one 8- / 16-byte load as floats (a bf16 is the upper half of an f32)
__device__ __forceinline__ float4 bf16x4_f32(uint2 w) {
    return make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                       __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
}

// DB: two LDS stages of TK keys; the next tile's K/V rows travel in registers while the current tile is multiplied and there
// is ONE barrier per tile (the shape that took the attention backward from 460 to 299 us, attn_bwd.hip).
// TR (round 3): the training forward - log-sum-exp output and dropout on the probabilities - so that sola_forward_train[_ragged]
// takes this shape too (its q-blocks are independent blocks: units of very different lengths balance over the chip, where
// attn.hip's shared mode walks a ragged unit's q-blocks inside one block).  A separate instantiation: the inference kernel's
// registers (126 at four blocks per CU) are untouched.
// IN16 (round 6, the training step's bf16 storage): q / k / v are BFLOAT16 rows (pitches in values) - 8-byte q loads, 16-byte K / V
// prefetch pieces of eight values, widened on their way into the f32 LDS stages; everything behind the loads is the f32 kernel.
template <int DH, int TK, bool DB = false, bool TR = false, bool IN16 = false>
__global__ __launch_bounds__(256, 4) void attn_fwd_f32_simple_kernel(const AttnSArgs a) {
    static_assert(!IN16 || (DB && TR), "bf16 inputs: the double-buffered training instantiation");
    constexpr int NC = DH / 16;
    constexpr int LDK = DH + 4, LDV = DH + 4;
    constexpr int F4 = DH / 4;
    extern __shared__ __attribute__((aligned(16))) float smem_s[];
    float* Ks = smem_s;
    float* Vs = smem_s + TK * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    // a.xcd_remap (grid % 8 == 0): consecutive logical blocks - the heads of a group, the q-blocks of a unit - share an XCD's L2
    const unsigned lb = a.xcd_remap ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const long long unit = lb / a.nqb;
    const int qb = (int)(lb - unit * a.nqb);
    const int grp = (int)(unit / a.H), h = (int)(unit - (long long)grp * a.H);
    long long q0, k0, q_rs, k_rs;
    int Sq, Sk;
    if (a.q_units) {
        const int4 qu = a.q_units[grp], ku = a.k_units[grp];
        q0 = __builtin_amdgcn_readfirstlane(qu.x); q_rs = __builtin_amdgcn_readfirstlane(qu.y); Sq = __builtin_amdgcn_readfirstlane(qu.z);
        k0 = __builtin_amdgcn_readfirstlane(ku.x); k_rs = __builtin_amdgcn_readfirstlane(ku.y); Sk = __builtin_amdgcn_readfirstlane(ku.z);
    } else {
        q0 = (long long)(grp / a.inner) * a.q_outer + (long long)(grp % a.inner) * a.q_inner;
        k0 = (long long)(grp / a.inner) * a.k_outer + (long long)(grp % a.inner) * a.k_inner;
        q_rs = a.q_rs; k_rs = a.k_rs; Sq = a.Sq; Sk = a.Sk;
    }
    if (qb * 64 >= Sq) return;  // ragged: fewer q-blocks than the largest unit (block-uniform)
    const int qi = qb * 64 + wave * 16 + c16;
    const bool q_ok = qi < Sq;
    float4 qf[NC];
    if constexpr (IN16) {
        const unsigned short* qp = reinterpret_cast<const unsigned short*>(a.q) + (q0 + (long long)(q_ok ? qi : 0) * q_rs) * a.ldq + h * DH + 4 * g4;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 v = bf16x4_f32(*reinterpret_cast<const uint2*>(qp + c * 16));
            qf[c] = q_ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        const float* qp = a.q + (q0 + (long long)qi * q_rs) * a.ldq + h * DH + 4 * g4;
#pragma unroll
        for (int c = 0; c < NC; ++c) qf[c] = q_ok ? *reinterpret_cast<const float4*>(qp + c * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4 oacc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    // pieces per thread and tile (K + V) of the prefetch: 16 bytes each - four f32 values, or eight bf16 ones (IN16)
    constexpr int VPP = IN16 ? 8 : 4, PPR = DH / VPP;  // values per piece, pieces per row
    constexpr int PT = DB ? 2 * TK * PPR / 256 : 1;
    float4 st[PT];
    // round 5: this kernel's time is its SIMD issue time - PMC: the f32 MFMA busy 50 % of the SIMD cycles and 1 066 VALU instructions per
    // wave, which do not overlap it on gfx950 (profiles/r05_attention_ring.txt) - so the prefetch keeps a pointer per piece and advances it
    // by the tile's stride (the 64-bit row products, quarter-rate multiplies, left the loop) and whole tiles load without the row test
    const char* fp[PT];
    int fr[PT];
    constexpr int ES = IN16 ? 2 : 4;  // bytes per stored value
    if constexpr (DB) {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int e = tid + 256 * j, which = e / (TK * PPR), r = (e % (TK * PPR)) / PPR, cp = e % PPR;
            fr[j] = r;
            fp[j] = reinterpret_cast<const char*>(which ? a.v : a.k) + ((k0 + (long long)r * k_rs) * (which ? a.ldv : a.ldk) + h * DH + cp * VPP) * ES;
        }
    }
    const long long fstep_k = (long long)TK * k_rs * a.ldk * ES, fstep_v = (long long)TK * k_rs * a.ldv * ES;  // wave-uniform, bytes
    auto fetch = [&](int kt0) {
        if (kt0 + TK <= Sk) {  // a whole tile (block-uniform)
#pragma unroll
            for (int j = 0; j < PT; ++j) st[j] = *reinterpret_cast<const float4*>(fp[j]);
        } else {
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                st[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kt0 + fr[j] < Sk) st[j] = *reinterpret_cast<const float4*>(fp[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < PT; ++j) fp[j] += (tid + 256 * j) / (TK * PPR) ? fstep_v : fstep_k;
    };
    if (DB) fetch(0);
    int it = 0;
    for (int kt0 = 0; kt0 < Sk; kt0 += TK, ++it) {
        const int nrows = min(TK, Sk - kt0);
        const int nrows16 = (nrows + 15) & ~15;
        if (DB) {
            Ks = smem_s + (it & 1) * 2 * TK * LDK;
            Vs = Ks + TK * LDK;
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const int e = tid + 256 * j, which = e / (TK * PPR), r = (e % (TK * PPR)) / PPR, cp = e % PPR;
                float* dst = &(which ? Vs : Ks)[r * LDK + cp * VPP];
                if constexpr (IN16) {
                    const uint4 w = __builtin_bit_cast(uint4, st[j]);
                    *reinterpret_cast<float4*>(dst) = bf16x4_f32(make_uint2(w.x, w.y));
                    *reinterpret_cast<float4*>(dst + 4) = bf16x4_f32(make_uint2(w.z, w.w));
                } else {
                    *reinterpret_cast<float4*>(dst) = st[j];
                }
            }
            __syncthreads();
            if (kt0 + TK < Sk) fetch(kt0 + TK);
        } else {
        if (kt0 > 0) __syncthreads();
        for (int idx = tid; idx < nrows16 * F4; idx += 256) {
            const int r = idx / F4, c4 = idx - r * F4;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (r < nrows) {
                const long long row = k0 + (long long)(kt0 + r) * k_rs;
                kv = *reinterpret_cast<const float4*>(a.k + row * a.ldk + h * DH + c4 * 4);
                vv = *reinterpret_cast<const float4*>(a.v + row * a.ldv + h * DH + c4 * 4);
            }
            *reinterpret_cast<float4*>(&Ks[r * LDK + c4 * 4]) = kv;
            *reinterpret_cast<float4*>(&Vs[r * LDV + c4 * 4]) = vv;
        }
        __syncthreads();
        }
        const int ntile = nrows16 >> 4;
        f32x4 sc[TK / 16];
#pragma unroll
        for (int t = 0; t < TK / 16; ++t) {
            if (t < ntile) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                const float* kp = &Ks[(t * 16 + c16) * LDK + 4 * g4];
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
                for (int r = 0; r < 4; ++r) sc[t][r] = (key0 + r < Sk) ? (a0[r] + a1[r]) * a.scale : -INFINITY;
            } else {
                sc[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < TK / 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
        mx = max_xor32(max_xor16(mx));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < TK / 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[t][r] = __expf(sc[t][r] - m_new);
                rs += sc[t][r];
            }
        rs = sum_xor32(sum_xor16(rs));
        l_run = l_run * alpha + rs;
        m_run = m_new;
        // the accumulators are rescaled only where some row's running maximum moved (x * 1 = x: the same bits; 32 multiplies of the SIMD's
        // issue time saved on every other tile or so)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) oacc[c] *= alpha;
        }
        if constexpr (TR) {
            if (a.drop.enabled) {  // dropout acts on the normalised probabilities: the row sum above stays undropped (attn.hip)
                const unsigned long long rbase = ((unsigned long long)(grp * a.H + h) * Sq + qi) * Sk;
#pragma unroll
                for (int t = 0; t < TK / 16; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sc[t][r] = dropout_keep(a.drop, rbase + kt0 + t * 16 + 4 * g4 + r) ? sc[t][r] * a.drop.scale : 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < TK / 16; ++t) {
            if (t < ntile) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* vp = &Vs[(t * 16 + 4 * g4 + r) * LDV + c16];
#pragma unroll
                    for (int c = 0; c < NC; ++c) oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[c * 16], sc[t][r], oacc[c], 0, 0, 0);
                }
            }
        }
    }
    if (!q_ok) return;
    const float inv = 1.f / l_run;
    if constexpr (TR) {
        if (a.lse && g4 == 0) a.lse[(q0 + (long long)qi * q_rs) * a.H + h] = m_run + logf(l_run);
    }
    float* op = a.o + (q0 + (long long)qi * q_rs) * a.ldo + h * DH;
    if (!a.o_sp16) {
        if (!(TR && a.o_skip_f32)) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                *reinterpret_cast<float4*>(op + 4 * g4 + c * 16) = make_float4(oacc[c][0] * inv, oacc[c][1] * inv, oacc[c][2] * inv, oacc[c][3] * inv);
        }
        if constexpr (TR) {
            if (a.o_cast) {  // the out-projection's operand cast of the same values (uniform branch)
                const long long eo = (q0 + (long long)qi * q_rs) * a.ldo + h * DH;  // element offset of this row's head slice
                if (a.o_cast_fmt == 1) {
                    float* pp = static_cast<float*>(a.o_cast) + eo;
                    _Float16* sd = a.o_side ? static_cast<_Float16*>(a.o_side) + eo : nullptr;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        half4v hi, lo;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            _Float16 h1, l1;
                            split_f16(oacc[c][j] * inv, h1, l1);
                            hi[j] = h1; lo[j] = l1;
                        }
                        char* dst = reinterpret_cast<char*>(pp + c * 16 + 8 * (g4 >> 1)) + 8 * (g4 & 1);
                        *reinterpret_cast<half4v*>(dst) = hi;
                        *reinterpret_cast<half4v*>(dst + 16) = lo;
                        if (sd) *reinterpret_cast<half4v*>(sd + c * 16 + 4 * g4) = hi;
                    }
                } else {
                    _Float16* cp = static_cast<_Float16*>(a.o_cast) + eo;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        half4v hv;
                        if (a.o_cast_fmt == 3) {
                            typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
                            bf16x4s b;
#pragma unroll
                            for (int j = 0; j < 4; ++j) b[j] = (__bf16)(oacc[c][j] * inv);
                            hv = __builtin_bit_cast(half4v, b);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) hv[j] = (_Float16)(oacc[c][j] * inv);
                        }
                        *reinterpret_cast<half4v*>(cp + c * 16 + 4 * g4) = hv;
                    }
                }
            }
        }
        return;
    }
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        half4v hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = oacc[c][j] * inv;
            _Float16 h1, l1;
            split_f16(v, h1, l1);
            hi[j] = h1; lo[j] = l1;
            m = fmaxf(m, fabsf(v));
        }
        char* dst = reinterpret_cast<char*>(op + c * 16 + 8 * (g4 >> 1)) + 8 * (g4 & 1);
        *reinterpret_cast<half4v*>(dst) = hi;
        *reinterpret_cast<half4v*>(dst + 16) = lo;
    }
    if (a.guard && !(m < 65000.f)) atomicOr(a.guard, 1);
}

// ---- sequences of at most 4 steps (motion attention over T' = ceil(T/8): 4 at the headline shape) -----------------------
// A (track, head) unit is 4 rows of q, k, v in and 4 rows of o out - 8 KB of traffic for 4 K multiply-adds: pure streaming.
// The packed MFMA shape of attn.hip stages K/V through 67 KB of LDS (two blocks per CU) and reaches 45 % of the HBM peak
// where a copy with this access pattern reaches 66-74 % (tools/micro/strided_bw).  Here half a wave owns a unit: lane c holds
// float4 chunk c of every row (rows are 32 chunks = one 512-byte head slice, so every load and store is a coalesced 512
// bytes), the TT x TT scores are per-lane partial dot products all-reduced over the 32 lanes, softmax and P V run in
// registers.  No LDS, ~100 VGPRs, any occupancy the registers allow.
template <int TT>
__global__ __launch_bounds__(256) void attn_fwd_small_kernel(const AttnSArgs a) {
    constexpr int DH = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, hw = lane >> 5;
    const long long unit = ((long long)blockIdx.x * 4 + wave) * 2 + hw;
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
    float4 qv[TT], kv[TT], vv[TT];
    // Rows past the unit's length: load a CLAMPED row and zero the VALUE.  `t < Sq ? *p : z` made the compiler select between the
    // global pointer and the address of a zero constant it had put into scratch memory, then load through a FLAT address: a
    // 16-byte scratch store per lane and row (366 MiB of WRITE_SIZE per launch for 256 MiB of output, profiles/r02_summary.md -
    // WRITE_SIZE itself is exact on this access pattern: profiles/r03_write_size_calibration.txt) and flat instead of global loads.
    auto ld = [&](const float* base, long long row0, long long rs, int ld_, int t, int n) {
        const float4 v = *reinterpret_cast<const float4*>(base + (row0 + (long long)(t < n ? t : 0) * rs) * ld_ + h * DH + 4 * c);
        const bool ok = t < n;
        return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    };
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        qv[t] = ld(a.q, q0, q_rs, a.ldq, t, Sq);
        kv[t] = ld(a.k, k0, k_rs, a.ldk, t, Sk);
        vv[t] = ld(a.v, k0, k_rs, a.ldv, t, Sk);
    }
    float sc[TT][TT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const float p = (qv[i].x * kv[j].x + qv[i].y * kv[j].y) + (qv[i].z * kv[j].z + qv[i].w * kv[j].w);
            sc[i][j] = half_sum32(p) * a.scale;  // DPP + one permlane swap; stays inside the 32-lane half
        }
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
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const float w = pj[j] * inv;
            o.x += w * vv[j].x; o.y += w * vv[j].y; o.z += w * vv[j].z; o.w += w * vv[j].w;
        }
        float* op = a.o + (q0 + (long long)i * q_rs) * a.ldo + h * DH;
        if (!a.o_sp16) {
            *reinterpret_cast<float4*>(op + 4 * c) = o;
        } else {
            // floats 4c .. 4c+3 are one half of the 8-wide block c >> 1 = 32 bytes [hi8 | lo8].  The lane pair (c, c ^ 1) trades
            // halves (two DPP moves each way) so that the even lane writes the 16 bytes of hi8 and the odd lane the 16 bytes of
            // lo8 - one 16-byte store per lane instead of two 8-byte ones (which reached HBM as 366 MiB of WRITE_SIZE per launch
            // for 256 MiB of output: half-written 32-byte sectors)
            const float v4[4] = {o.x, o.y, o.z, o.w};
            half4v hi, lo;
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 h1, l1;
                split_f16(v4[j], h1, l1);
                hi[j] = h1; lo[j] = l1;
                m = fmaxf(m, fabsf(v4[j]));
            }
            const int2 hw2 = __builtin_bit_cast(int2, hi), lw2 = __builtin_bit_cast(int2, lo);
            const bool odd = c & 1;
            const int sx = odd ? hw2.x : lw2.x, sy = odd ? hw2.y : lw2.y;  // what the partner needs: its lo8 needs my lo4, its hi8 my hi4
            const int rx = __builtin_amdgcn_update_dpp(0, sx, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]: lane c <- lane c ^ 1
            const int ry = __builtin_amdgcn_update_dpp(0, sy, 0xB1, 0xF, 0xF, false);
            // even lane: [my hi4 | partner's hi4] at block + 0;  odd lane: [partner's lo4 | my lo4] at block + 16
            const int4 piece = odd ? make_int4(rx, ry, lw2.x, lw2.y) : make_int4(hw2.x, hw2.y, rx, ry);
            *reinterpret_cast<int4*>(reinterpret_cast<char*>(op + 8 * (c >> 1)) + (odd ? 16 : 0)) = piece;
            if (a.guard && !(m < 65000.f)) atomicOr(a.guard, 1);
        }
    }
}

// ---- split-f16 products on f32 inputs (the split precision mode) -----------------------------------------------------------
// The exact-f32 MFMA (v_mfma_f32_16x16x4_f32: 32 cycles for 2 K flops) makes the shapes above co-bound by the matrix pipe:
// the object -> language attention needs 94 us of it per launch next to 114 us of HBM time at the practical rate, and the two
// do not overlap well at two to four waves per SIMD (measured 204 us = their sum).  In the split precision mode every other
// contraction already runs as hi*hi + hi*lo + lo*hi on f16 MFMAs (22-bit products), and so can these: the K/V tile is
// converted to (hi, lo) halfs while it is staged - K row-major, V TRANSPOSED ([d][key]) so that both MFMA A-fragments are
// single 8-byte LDS reads - Q and P are split in registers, and a product costs 3 x 8 cycles instead of 4 x 32.  The halved
// tile (38 KB for 32 keys) keeps four blocks per CU.  Same block decomposition, addressing and online softmax as above.
__device__ __forceinline__ void split4h(const float4 v, half4v& hi, half4v& lo) {
    const float in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        _Float16 h1, l1;
        split_f16(in[j], h1, l1);
        hi[j] = h1; lo[j] = l1;
    }
}
__device__ __forceinline__ f32x4 mfma3h(const half4v ah, const half4v al, const half4v bh, const half4v bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c, 0, 0, 0);
}

template <int DH, int TK>
__global__ __launch_bounds__(256, 4) void attn_fwd_splitm_kernel(const AttnSArgs a) {
    constexpr int NC = DH / 16;
    constexpr int LDK = DH + 8;   // halfs: K rows, 8-byte fragment reads conflict-free (row pitch = 4 banks mod 64)
    constexpr int LDT = TK + 8;   // halfs: V^T rows (one per head dim), pitch 20 dwords at TK = 32
    constexpr int F4 = DH / 4;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_m[];
    _Float16* Kh = smem_m;
    _Float16* Kl = Kh + TK * LDK;
    _Float16* Vh = Kl + TK * LDK;   // [DH][LDT]
    _Float16* Vl = Vh + DH * LDT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    const long long unit = blockIdx.x / a.nqb;
    const int qb = blockIdx.x - (int)unit * a.nqb;
    const int grp = (int)(unit / a.H), h = (int)(unit - (long long)grp * a.H);
    long long q0, k0, q_rs, k_rs;
    int Sq, Sk;
    if (a.q_units) {
        const int4 qu = a.q_units[grp], ku = a.k_units[grp];
        q0 = __builtin_amdgcn_readfirstlane(qu.x); q_rs = __builtin_amdgcn_readfirstlane(qu.y); Sq = __builtin_amdgcn_readfirstlane(qu.z);
        k0 = __builtin_amdgcn_readfirstlane(ku.x); k_rs = __builtin_amdgcn_readfirstlane(ku.y); Sk = __builtin_amdgcn_readfirstlane(ku.z);
    } else {
        q0 = (long long)(grp / a.inner) * a.q_outer + (long long)(grp % a.inner) * a.q_inner;
        k0 = (long long)(grp / a.inner) * a.k_outer + (long long)(grp % a.inner) * a.k_inner;
        q_rs = a.q_rs; k_rs = a.k_rs; Sq = a.Sq; Sk = a.Sk;
    }
    if (qb * 64 >= Sq) return;
    const int qi = qb * 64 + wave * 16 + c16;
    const bool q_ok = qi < Sq;
    half4v qh[NC], ql[NC];
    {
        const float* qp = a.q + (q0 + (long long)qi * q_rs) * a.ldq + h * DH + 4 * g4;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 v = q_ok ? *reinterpret_cast<const float4*>(qp + c * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            split4h(v, qh[c], ql[c]);
        }
    }
    f32x4 oacc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    for (int kt0 = 0; kt0 < Sk; kt0 += TK) {
        const int nrows = min(TK, Sk - kt0);
        const int nrows16 = (nrows + 15) & ~15;
        if (kt0 > 0) __syncthreads();
        for (int idx = tid; idx < nrows16 * F4; idx += 256) {
            const int r = idx / F4, c4 = idx - r * F4;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (r < nrows) {
                const long long row = k0 + (long long)(kt0 + r) * k_rs;
                kv = *reinterpret_cast<const float4*>(a.k + row * a.ldk + h * DH + c4 * 4);
                vv = *reinterpret_cast<const float4*>(a.v + row * a.ldv + h * DH + c4 * 4);
            }
            half4v hi, lo;
            split4h(kv, hi, lo);
            *reinterpret_cast<half4v*>(&Kh[r * LDK + c4 * 4]) = hi;
            *reinterpret_cast<half4v*>(&Kl[r * LDK + c4 * 4]) = lo;
            split4h(vv, hi, lo);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Vh[(c4 * 4 + j) * LDT + r] = hi[j];
                Vl[(c4 * 4 + j) * LDT + r] = lo[j];
            }
        }
        __syncthreads();
        const int ntile = nrows16 >> 4;
        f32x4 sc[TK / 16];
#pragma unroll
        for (int t = 0; t < TK / 16; ++t) {
            if (t < ntile) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                const int ko = (t * 16 + c16) * LDK + 4 * g4;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const half4v kh = *reinterpret_cast<const half4v*>(&Kh[ko + c * 16]);
                    const half4v kl = *reinterpret_cast<const half4v*>(&Kl[ko + c * 16]);
                    if (c & 1) a1 = mfma3h(kh, kl, qh[c], ql[c], a1);
                    else a0 = mfma3h(kh, kl, qh[c], ql[c], a0);
                }
                const int key0 = kt0 + t * 16 + 4 * g4;
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[t][r] = (key0 + r < Sk) ? (a0[r] + a1[r]) * a.scale : -INFINITY;
            } else {
                sc[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < TK / 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
        mx = max_xor32(max_xor16(mx));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < TK / 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[t][r] = __expf(sc[t][r] - m_new);
                rs += sc[t][r];
            }
        rs = sum_xor32(sum_xor16(rs));
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int c = 0; c < NC; ++c) oacc[c] *= alpha;
#pragma unroll
        for (int t = 0; t < TK / 16; ++t) {
            if (t < ntile) {
                half4v ph, pl;
                split4h(make_float4(sc[t][0], sc[t][1], sc[t][2], sc[t][3]), ph, pl);
                const int vo = c16 * LDT + t * 16 + 4 * g4;  // V^T row d = 16c + c16, keys 16t + 4*g4 .. +3
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const half4v vh = *reinterpret_cast<const half4v*>(&Vh[vo + c * 16 * LDT]);
                    const half4v vl = *reinterpret_cast<const half4v*>(&Vl[vo + c * 16 * LDT]);
                    oacc[c] = mfma3h(vh, vl, ph, pl, oacc[c]);
                }
            }
        }
    }
    if (!q_ok) return;
    const float inv = 1.f / l_run;
    float* op = a.o + (q0 + (long long)qi * q_rs) * a.ldo + h * DH;
    if (!a.o_sp16) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            *reinterpret_cast<float4*>(op + 4 * g4 + c * 16) = make_float4(oacc[c][0] * inv, oacc[c][1] * inv, oacc[c][2] * inv, oacc[c][3] * inv);
        return;
    }
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        half4v hi, lo;
        const float4 v = make_float4(oacc[c][0] * inv, oacc[c][1] * inv, oacc[c][2] * inv, oacc[c][3] * inv);
        split4h(v, hi, lo);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        char* dst = reinterpret_cast<char*>(op + c * 16 + 8 * (g4 >> 1)) + 8 * (g4 & 1);
        *reinterpret_cast<half4v*>(dst) = hi;
        *reinterpret_cast<half4v*>(dst + 16) = lo;
    }
    if (a.guard && !(m < 65000.f)) atomicOr(a.guard, 1);
}

// ---- split-f16 q / k / v (written as split pairs by the projection GEMMs, GemmDesc::c_sp16), high-occupancy shape (round 3) -----
// From ~48 keys per unit on, the exact-f32 products above take as long at the matrix pipe's peak as the traffic takes at the HBM
// peak (object -> language at the headline shape: 82 us against 115 us; inter-object attention over 128 tracks: 219 against 134),
// and the two do not overlap at four waves per SIMD.  attn.hip's kernel for split inputs is the round-1 shape (8 waves, 68 KB of
// LDS: 401 us where the f32 shape above takes 280-324).  Here the split rows go through the SAME block decomposition as the
// f32 shape: a row of 128 values is 16 blocks of [hi8 | lo8] = 512 bytes, K / V tiles are staged with plain 16-byte copies
// (no conversion: the GEMM epilogue did it), and
//   S^T = K Q^T : A = the key's (hi4, lo4) of a 16-wide head-dim chunk - two 8-byte LDS reads - B = Q's, loaded once;
//   O^T = V^T P^T: A = V^T fragments through ds_read_b64_tr_b16 (the 4 x 16 block of four keys x sixteen head dims, each lane
//                 supplying the address of one 8-byte piece of the split row: no transposed copy of V anywhere), B = P split
//                 in registers;
// each product hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x16_f16: 3 x 16 cycles where the f32 shape spends 4 x 32.
// Row pitch 560 bytes: both read patterns conflict-free on the 64-bank LDS.
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));
// DB: two LDS stages of TK keys, the next tile's rows travel in registers under the current tile's products, one barrier per tile
// (as the f32 shape above).
template <int TK, bool DB = false>
__global__ __launch_bounds__(256, 4) void attn_fwd_spin_kernel(const AttnSArgs a) {
    constexpr int DH = 128, NC = DH / 16;
    constexpr int RB = 560;  // LDS row pitch, bytes (512 + 48)
    constexpr int P16 = DH / 4;  // 16-byte pieces per split row (DH values x 4 bytes)
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    char* Ks = smem_p;
    char* Vs = smem_p + TK * RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    const long long unit = blockIdx.x / a.nqb;
    const int qb = blockIdx.x - (int)unit * a.nqb;
    const int grp = (int)(unit / a.H), h = (int)(unit - (long long)grp * a.H);
    long long q0, k0, q_rs, k_rs;
    int Sq, Sk;
    if (a.q_units) {
        const int4 qu = a.q_units[grp], ku = a.k_units[grp];
        q0 = __builtin_amdgcn_readfirstlane(qu.x); q_rs = __builtin_amdgcn_readfirstlane(qu.y); Sq = __builtin_amdgcn_readfirstlane(qu.z);
        k0 = __builtin_amdgcn_readfirstlane(ku.x); k_rs = __builtin_amdgcn_readfirstlane(ku.y); Sk = __builtin_amdgcn_readfirstlane(ku.z);
    } else {
        q0 = (long long)(grp / a.inner) * a.q_outer + (long long)(grp % a.inner) * a.q_inner;
        k0 = (long long)(grp / a.inner) * a.k_outer + (long long)(grp % a.inner) * a.k_inner;
        q_rs = a.q_rs; k_rs = a.k_rs; Sq = a.Sq; Sk = a.Sk;
    }
    if (qb * 64 >= Sq) return;
    const int qi = qb * 64 + wave * 16 + c16;
    const bool q_ok = qi < Sq;
    // byte offset, inside a split row, of the hi4 this lane's 4 head dims of chunk 0 (16c + 4*g4 .. +3); lo4 sits 16 bytes on
    const int frag = (g4 >> 1) * 32 + (g4 & 1) * 8;
    half4v qh[NC], ql[NC];
    {
        const char* qp = reinterpret_cast<const char*>(a.q + (q0 + (long long)(q_ok ? qi : 0) * q_rs) * a.ldq + h * DH) + frag;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const half4v hv = *reinterpret_cast<const half4v*>(qp + c * 64);
            const half4v lv = *reinterpret_cast<const half4v*>(qp + c * 64 + 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qh[c][j] = q_ok ? hv[j] : (_Float16)0.f;
                ql[c][j] = q_ok ? lv[j] : (_Float16)0.f;
            }
        }
    }
    f32x4 oacc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    // V^T fragment addresses (ds_read_b64_tr_b16): lane p of a 16-lane group supplies the piece (key 4*g4 + (p >> 2), head dims
    // 4 * (p & 3) .. +3 of the chunk) and receives head dim p of the four keys
    const int vfrag = (4 * g4 + (c16 >> 2)) * RB + ((c16 & 3) >> 1) * 32 + (c16 & 1) * 8;
    constexpr int PT = DB ? 2 * TK * P16 / 256 : 1;  // 16-byte pieces per thread and tile (K + V) of the prefetch
    float4 st[PT];
    auto fetch = [&](int kt0) {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int e = tid + 256 * j, which = e / (TK * P16), r = (e % (TK * P16)) / P16, c4 = e % P16;
            st[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kt0 + r < Sk) {
                const long long row = k0 + (long long)(kt0 + r) * k_rs;
                st[j] = which ? *reinterpret_cast<const float4*>(a.v + row * a.ldv + h * DH + c4 * 4)
                              : *reinterpret_cast<const float4*>(a.k + row * a.ldk + h * DH + c4 * 4);
            }
        }
    };
    if (DB) fetch(0);
    int it = 0;
    for (int kt0 = 0; kt0 < Sk; kt0 += TK, ++it) {
        const int nrows = min(TK, Sk - kt0);
        const int nrows16 = (nrows + 15) & ~15;
        if (DB) {
            Ks = smem_p + (it & 1) * 2 * TK * RB;
            Vs = Ks + TK * RB;
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const int e = tid + 256 * j, which = e / (TK * P16), r = (e % (TK * P16)) / P16, c4 = e % P16;
                *reinterpret_cast<float4*>((which ? Vs : Ks) + r * RB + c4 * 16) = st[j];
            }
            __syncthreads();
            if (kt0 + TK < Sk) fetch(kt0 + TK);
        } else {
        if (kt0 > 0) __syncthreads();
        for (int idx = tid; idx < nrows16 * P16; idx += 256) {
            const int r = idx / P16, c4 = idx - r * P16;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (r < nrows) {
                const long long row = k0 + (long long)(kt0 + r) * k_rs;
                kv = *reinterpret_cast<const float4*>(a.k + row * a.ldk + h * DH + c4 * 4);
                vv = *reinterpret_cast<const float4*>(a.v + row * a.ldv + h * DH + c4 * 4);
            }
            *reinterpret_cast<float4*>(Ks + r * RB + c4 * 16) = kv;
            *reinterpret_cast<float4*>(Vs + r * RB + c4 * 16) = vv;
        }
        __syncthreads();
        }
        const int ntile = nrows16 >> 4;
        f32x4 sc[TK / 16];
#pragma unroll
        for (int t = 0; t < TK / 16; ++t) {
            if (t < ntile) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                const char* kp = Ks + (t * 16 + c16) * RB + frag;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const half4v kh = *reinterpret_cast<const half4v*>(kp + c * 64);
                    const half4v kl = *reinterpret_cast<const half4v*>(kp + c * 64 + 16);
                    if (c & 1) a1 = mfma3h(kh, kl, qh[c], ql[c], a1);
                    else a0 = mfma3h(kh, kl, qh[c], ql[c], a0);
                }
                const int key0 = kt0 + t * 16 + 4 * g4;
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[t][r] = (key0 + r < Sk) ? (a0[r] + a1[r]) * a.scale : -INFINITY;
            } else {
                sc[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < TK / 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
        mx = max_xor32(max_xor16(mx));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < TK / 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[t][r] = __expf(sc[t][r] - m_new);
                rs += sc[t][r];
            }
        rs = sum_xor32(sum_xor16(rs));
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int c = 0; c < NC; ++c) oacc[c] *= alpha;
#pragma unroll
        for (int t = 0; t < TK / 16; ++t) {
            if (t < ntile) {
                half4v ph, pl;
                split4h(make_float4(sc[t][0], sc[t][1], sc[t][2], sc[t][3]), ph, pl);
                const char* vp = Vs + t * 16 * RB + vfrag;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    typedef __attribute__((address_space(3))) short4v lds_s4;
                    const short4v vh4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(vp + c * 64));
                    const short4v vl4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(vp + c * 64 + 16));
                    oacc[c] = mfma3h(__builtin_bit_cast(half4v, vh4), __builtin_bit_cast(half4v, vl4), ph, pl, oacc[c]);
                }
            }
        }
    }
    if (!q_ok) return;
    const float inv = 1.f / l_run;
    float* op = a.o + (q0 + (long long)qi * q_rs) * a.ldo + h * DH;
    if (!a.o_sp16) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            *reinterpret_cast<float4*>(op + 4 * g4 + c * 16) = make_float4(oacc[c][0] * inv, oacc[c][1] * inv, oacc[c][2] * inv, oacc[c][3] * inv);
        return;
    }
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        half4v hi, lo;
        const float4 v = make_float4(oacc[c][0] * inv, oacc[c][1] * inv, oacc[c][2] * inv, oacc[c][3] * inv);
        split4h(v, hi, lo);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        char* dst = reinterpret_cast<char*>(op + c * 16 + 8 * (g4 >> 1)) + 8 * (g4 & 1);
        *reinterpret_cast<half4v*>(dst) = hi;
        *reinterpret_cast<half4v*>(dst + 16) = lo;
    }
    if (a.guard && !(m < 65000.f)) atomicOr(a.guard, 1);
}

int g_attn_spin_db = 1;  // sola_tune "attn_spin" 2 = single-buffered 32-key stages (A/B)
int launch_spin(const AttnSArgs& a0, hipStream_t s) {
    AttnSArgs a = a0;
    constexpr int TK = 32;
    a.nqb = (a.Sq + 63) / 64;
    const long long blocks = (long long)a.G * a.H * a.nqb;
    SOLA_ARG(blocks < (1ll << 31), "attention: grid too large");
    if (g_attn_spin_db) {  // two stages of 16 keys (the same 35.8 KB), next tile prefetched in registers
        hipLaunchKernelGGL((attn_fwd_spin_kernel<16, true>), dim3((unsigned)blocks), dim3(256), (size_t)2 * 2 * 16 * 560, s, a);
        SOLA_LAUNCH_CHECK();
        return SOLA_OK;
    }
    const size_t lds = (size_t)2 * TK * 560;
    hipLaunchKernelGGL((attn_fwd_spin_kernel<TK>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_splitm(const AttnSArgs& a0, hipStream_t s) {
    AttnSArgs a = a0;
    constexpr int DH = 128, TK = 32;
    a.nqb = (a.Sq + 63) / 64;
    const long long blocks = (long long)a.G * a.H * a.nqb;
    SOLA_ARG(blocks < (1ll << 31), "attention: grid too large");
    const size_t lds = ((size_t)2 * TK * (DH + 8) + (size_t)2 * DH * (TK + 8)) * sizeof(_Float16);
    hipLaunchKernelGGL((attn_fwd_splitm_kernel<DH, TK>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int g_attn_simple_db = 1;  // sola_tune "attn_simple_db": 1 = double-buffered 16-key stages with register prefetch
int g_attn_simple_remap = 0;  // sola_tune "attn_simple_remap": XCD-contiguous block order; measured no gain (273.8 vs 275.8 us at 64 tracks, 392 vs 400 at 128)

template <int DH>
int launch_s(const AttnSArgs& a0, hipStream_t s) {
    AttnSArgs a = a0;
    constexpr int TK = 32;
    a.nqb = (a.Sq + 63) / 64;
    const long long blocks = (long long)a.G * a.H * a.nqb;
    SOLA_ARG(blocks < (1ll << 31), "attention: grid too large");
    a.xcd_remap = (g_attn_simple_remap && blocks % 8 == 0) ? 1 : 0;
    if (a.in_bf16) {  // training forward on bf16 q / k / v
        const size_t lds2 = (size_t)2 * 2 * 16 * (DH + 4) * sizeof(float);
        hipLaunchKernelGGL((attn_fwd_f32_simple_kernel<DH, 16, true, true, true>), dim3((unsigned)blocks), dim3(256), lds2, s, a);
        SOLA_LAUNCH_CHECK();
        return SOLA_OK;
    }
    if (a.lse || a.drop.enabled) {  // training forward
        const size_t lds2 = (size_t)2 * 2 * 16 * (DH + 4) * sizeof(float);
        hipLaunchKernelGGL((attn_fwd_f32_simple_kernel<DH, 16, true, true>), dim3((unsigned)blocks), dim3(256), lds2, s, a);
        SOLA_LAUNCH_CHECK();
        return SOLA_OK;
    }
    if (g_attn_simple_db) {  // two stages of 16 keys: the same LDS footprint (34 KB), next tile prefetched in registers
        const size_t lds2 = (size_t)2 * 2 * 16 * (DH + 4) * sizeof(float);
        hipLaunchKernelGGL((attn_fwd_f32_simple_kernel<DH, 16, true>), dim3((unsigned)blocks), dim3(256), lds2, s, a);
        SOLA_LAUNCH_CHECK();
        return SOLA_OK;
    }
    const size_t lds = (size_t)2 * TK * (DH + 4) * sizeof(float);
    hipLaunchKernelGGL((attn_fwd_f32_simple_kernel<DH, TK>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

}  // namespace

void sola_attn_set_simple_remap(int v) { g_attn_simple_remap = v; }
void sola_attn_set_simple_db(int v) { g_attn_simple_db = v; }

// sequences of <= 4 steps at head_dim 128, inference (no log-sum-exp, no dropout), f32 q / k / v
bool attention_small_supported(const AttnDesc& d) {
    return !d.lse && !d.drop.enabled && !d.in_sp16 && d.Sq <= 4 && d.Sk <= 4 && d.DH == 128;
}

static AttnSArgs make_sargs(const AttnDesc& d) {
    AttnSArgs a;
    a.q = d.q; a.k = d.k; a.v = d.v; a.o = d.o;
    a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo;
    a.G = d.G; a.H = d.H; a.Sq = d.Sq; a.Sk = d.Sk; a.inner = d.inner; a.nqb = 1;
    a.q_outer = d.q_outer; a.q_inner = d.q_inner; a.q_rs = d.q_rs;
    a.k_outer = d.k_outer; a.k_inner = d.k_inner; a.k_rs = d.k_rs;
    a.scale = d.scale; a.o_sp16 = d.o_sp16; a.guard = d.o_sp16 ? d.guard : nullptr;
    a.q_units = d.q_units; a.k_units = d.q_units ? (d.k_units ? d.k_units : d.q_units) : nullptr;
    a.xcd_remap = 0;
    a.lse = d.lse;
    a.drop = d.drop;
    a.o_cast = nullptr; a.o_side = nullptr; a.o_cast_fmt = 0;  // launch_attention_simple's training instantiation takes them
    a.in_bf16 = 0; a.o_skip_f32 = 0;
    return a;
}

int launch_attention_small(const AttnDesc& d, hipStream_t s) {
    const AttnSArgs a = make_sargs(d);
    const double elems = (double)d.G * d.H * d.DH;
    SolaProfScope prof(SOLA_PROF_ATTN, s, 4.0 * elems * d.Sq * d.Sk, 4.0 * elems * (2.0 * d.Sq + 2.0 * d.Sk));
    const long long units = (long long)d.G * d.H;
    const unsigned blocks = (unsigned)((units + 7) / 8);
    const int need = d.Sq > d.Sk ? d.Sq : d.Sk;
    if (need <= 1) hipLaunchKernelGGL((attn_fwd_small_kernel<1>), dim3(blocks), dim3(256), 0, s, a);
    else if (need <= 2) hipLaunchKernelGGL((attn_fwd_small_kernel<2>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_small_kernel<4>), dim3(blocks), dim3(256), 0, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

// f32 q / k / v, more than 16 keys or queries; with a log-sum-exp output / dropout (training) the TR instantiation, f32 output only
int g_attn_simple_train = 1;  // sola_tune "attn_simple_train": 0 = the training forward keeps attn.hip's kernels (A/B)
bool attention_simple_supported(const AttnDesc& d) {
    if ((d.lse || d.drop.enabled) && (!g_attn_simple_train || d.o_sp16)) return false;
    return !d.in_sp16 && (d.Sq > 16 || d.Sk > 16) && (d.DH == 128 || d.DH == 64);
}
// bfloat16 q / k / v (AttnDesc::in_bf16): this file's training instantiation takes them - f32 output or bf16 cast only, 8-value-aligned rows
bool attention_in_bf16_supported(const AttnDesc& d) {
    return g_attn_simple_train && !d.o_sp16 && !d.in_sp16 && !d.k_private && (d.Sq > 16 || d.Sk > 16) && (d.DH == 128 || d.DH == 64) &&
           d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 8 == 0;
}
void sola_attn_set_simple_train(int v) { g_attn_simple_train = v; }

int launch_attention_simple(const AttnDesc& d, hipStream_t s) {
    AttnSArgs a = make_sargs(d);
    if (d.o_cast && d.o_cast_done && (d.lse || d.drop.enabled || d.in_bf16) && !d.o_sp16 && d.ldo % 8 == 0) {  // the TR instantiation runs (launch_s)
        a.o_cast = d.o_cast; a.o_side = d.o_cast_fmt == 1 ? d.o_side : nullptr; a.o_cast_fmt = d.o_cast_fmt;
        *d.o_cast_done = true;
    }
    a.in_bf16 = d.in_bf16;
    if (d.in_bf16) {
        SOLA_ARG(attention_in_bf16_supported(d), "attention: bf16 q / k / v need the high-occupancy training shape (more than 16 queries or keys, head_dim 64 / 128, pitches %% 8 == 0)");
        a.o_skip_f32 = d.o == nullptr;
        SOLA_ARG(d.o || a.o_cast, "attention: no output");
    }
    const double elems = (double)d.G * d.H * d.DH;
    SolaProfScope prof(SOLA_PROF_ATTN, s, 4.0 * elems * d.Sq * d.Sk, 4.0 * elems * (2.0 * d.Sq + 2.0 * d.Sk));
    return d.DH == 128 ? launch_s<128>(a, s) : launch_s<64>(a, s);
}

// split-f16 q / k / v (GEMM outputs written as split pairs), head_dim 128, inference: the high-occupancy shape for split inputs
int g_attn_spin = 1;  // sola_tune "attn_spin": 0 = attn.hip's round-1 kernel for split inputs (A/B)
void sola_attn_set_spin(int v) { g_attn_spin = v != 0; g_attn_spin_db = v != 2; }
bool attention_spin_supported(const AttnDesc& d) {
    // units of at most two 64-query blocks (the inter-object attention): with many q-blocks per unit (object -> language) every
    // block re-stages the unit's K / V and attn.hip's resident-K/V loop wins (158 vs 184 us at the headline shape)
    return g_attn_spin && d.in_sp16 && !d.lse && !d.drop.enabled && d.DH == 128 && (d.Sq > 16 || d.Sk > 16) && (d.Sq <= 128 || d.q_units) &&
           d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0;
}
int launch_attention_spin(const AttnDesc& d, hipStream_t s) {
    const AttnSArgs a = make_sargs(d);
    const double elems = (double)d.G * d.H * d.DH;
    SolaProfScope prof(SOLA_PROF_ATTN, s, 4.0 * elems * d.Sq * d.Sk, 4.0 * elems * (2.0 * d.Sq + 2.0 * d.Sk));
    return launch_spin(a, s);
}

// f32 q / k / v with the products evaluated as split-f16 triples (AttnDesc::split_math; head_dim 128, inference)
bool attention_splitm_supported(const AttnDesc& d) {
    return d.split_math && !d.lse && !d.drop.enabled && !d.in_sp16 && (d.Sq > 4 || d.Sk > 4) && d.DH == 128;
}
int launch_attention_splitm(const AttnDesc& d, hipStream_t s) {
    const AttnSArgs a = make_sargs(d);
    const double elems = (double)d.G * d.H * d.DH;
    SolaProfScope prof(SOLA_PROF_ATTN, s, 4.0 * elems * d.Sq * d.Sk, 4.0 * elems * (2.0 * d.Sq + 2.0 * d.Sk));
    return launch_splitm(a, s);
}
