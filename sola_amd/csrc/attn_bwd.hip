// Backward of the attention core (tools/attention.py:66-72 under autograd), exact f32 on v_mfma_f32_16x16x4_f32.
// Two passes that each recompute the probabilities from the saved log-sum-exp (no S x S matrix is ever stored):
//   attn_bwd_dq_kernel  : one wave per 16-query tile walks the keys.   P^T = exp(K Q^T - lse); dP^T = V dO^T;
//                         dS^T = P^T o (dP^T - D); dQ^T += K^T dS^T.   Also writes D[q] = dO[q] . O[q].
//   attn_bwd_dkv_kernel : one wave per 16-key tile walks the queries.  P = exp(Q K^T - lse); dP = dO V^T;
//                         dS = P o (dP - D); dV^T += dO^T P; dK^T += Q^T dS.
// In both passes the reduction index of every MFMA sits in the (lane>>4, register) slots and the row that owns the
// softmax statistics sits in the 16-lane column (pass 1) or is fetched per register slot (pass 2), so - as in the
// forward kernel - probabilities never move across lanes.  Every wave stages the streamed side through its own LDS
// slice (16 rows at a time); all waves of a launch run the same trip count, so block barriers are uniform.
// Addressing is the forward kernel's (outer / inner / row strides), so no permuted copies exist in the backward either.
#include "kernels.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct AttnBwdArgs {
    const float *q, *k, *v, *o, *dout, *lse;
    float *dq, *dk, *dv, *dvec;  // dvec [q rows][H]
    int ldq, ldk, ldv, ldo;      // ldo: row pitch of o / dout
    int ld_dq, ld_dk, ld_dv;     // row pitches of the gradient outputs (they may be column slices of one matrix)
    int G, H, Sq, Sk, inner, ntile;
    long long q_outer, q_inner, q_rs;
    long long k_outer, k_inner, k_rs;
    float scale;
    DropoutCfg drop;
    // ragged batches (AttnBwdDesc::q_units): group g has q_units[g] = (first row, row stride, Sq_g, -), k_units[g] likewise; Sq / Sk
    // are then the LARGEST lengths: they size the grids and - in the per-wave kernels, whose four waves serve different units
    // between shared block barriers - the loop trip counts; rows past a unit's own length are masked
    const int4 *q_units, *k_units;
    // one-pass kernel, long query ranges (object -> language): a block serves qc_tiles 16-query tiles of its unit (0 = all of them) and
    // writes its dK / dV partial sums to part[slot][2][16 * NWU][128], slot = first row / (16 * qc_tiles) + unit + chunk (injective for
    // units of row stride 1 laid out in unit order); attn_bwd_part_reduce_kernel adds a unit's chunks in order
    int qc_tiles;
    float* part;
    // round 6 (the training step's bf16 storage; one-pass kernel only - check attention_bwd_bf16_supported): q, k, v are BFLOAT16 rows
    // (ldq / ldk / ldv in values) and dQ, dK, dV leave as bfloat16 rows dq16 / dk16 / dv16 (pitches ld_dq / ld_dk / ld_dv in values) - the
    // operand of the dW / dX GEMMs behind this launch, no f32 copy and no cast pass.  dq (f32, same pitch) is scratch then: units of more
    // keys than one key group holds accumulate their dQ there in f32 and only the last group writes the bf16 row.
    // (dk / dv carry the bfloat16 key / value gradient pointers in this mode - two fewer kernel arguments: the four-wave shape has no scalar
    // register to spare)
    int io16;
    unsigned short* dq16;
    // round 6 (with io16 and the bf16 products): dout is a BFLOAT16 matrix too (pitch ldo in values) - the out-projection's input-gradient
    // GEMM wrote bfloat16 rows (what autocast's linear backward returns); the products take the same rounded values either way
    int g16;
    int o16;  // ... and o too (pitch ldo in values): the forward kept only the bfloat16 rows the out-projection reads (with g16)
};
__device__ __forceinline__ float4 bf16x4_to_f32(uint2 w) {
    return make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                       __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 f32x4_to_bf16(float x, float y, float z, float w) {
    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
    bf4 b;
    b[0] = (__bf16)x; b[1] = (__bf16)y; b[2] = (__bf16)z; b[3] = (__bf16)w;
    return __builtin_bit_cast(uint2, b);
}

struct BwdGeo { long long q0, k0, q_rs, k_rs; int Sq, Sk; };
__device__ __forceinline__ BwdGeo bwd_geo(const AttnBwdArgs& a, int grp) {
    BwdGeo g;
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

// RAG (ragged batches): every wave stages its unit through its OWN LDS slice, so nothing is shared between the waves of a block:
// the block barriers become wave barriers (a wave's LDS instructions execute in order), a wave whose query tile lies past its
// unit's length leaves at once, and the key / query loops run to the unit's OWN length instead of the batch's largest.
template <bool RAG>
__device__ __forceinline__ void stage_sync() {
    if constexpr (RAG) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <int DH, bool RAG = false>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnBwdArgs a) {
    constexpr int NC = DH / 16, LD = DH + 4, F4 = DH / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    float* Ks = smem + wave * 2 * 16 * LD;
    float* Vs = Ks + 16 * LD;

    const long long unit = (long long)blockIdx.x * 4 + wave;
    const bool unit_ok = unit < (long long)a.G * a.H * a.ntile;
    const int qt = unit_ok ? (int)(unit % a.ntile) : 0;
    const long long gh = unit_ok ? unit / a.ntile : 0;
    const int h = (int)(gh % a.H), grp = (int)(gh / a.H);
    const BwdGeo geo = bwd_geo(a, grp);
    const long long qrow0 = geo.q0, krow0 = geo.k0;
    const int qi = qt * 16 + c16;
    const bool q_ok = unit_ok && qi < geo.Sq;
    if constexpr (RAG) {
        if (!unit_ok || qt * 16 >= geo.Sq) return;  // the whole wave (its tile is past the unit's queries)
    }
    const int sk_loop = RAG ? geo.Sk : a.Sk;
    const long long qrow = qrow0 + (long long)(q_ok ? qi : 0) * geo.q_rs;

    float4 qf[NC], dof[NC];
    float dsum = 0.f;
    {
        const float* qp = a.q + qrow * a.ldq + h * DH + 4 * g4;
        const float* op = a.o + qrow * a.ldo + h * DH + 4 * g4;
        const float* gp = a.dout + qrow * a.ldo + h * DH + 4 * g4;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 qv = q_ok ? *reinterpret_cast<const float4*>(qp + c * 16) : z;
            const float4 ov = q_ok ? *reinterpret_cast<const float4*>(op + c * 16) : z;
            const float4 gv = q_ok ? *reinterpret_cast<const float4*>(gp + c * 16) : z;
            qf[c] = make_float4(qv.x * a.scale, qv.y * a.scale, qv.z * a.scale, qv.w * a.scale);
            dof[c] = gv;
            dsum += (ov.x * gv.x + ov.y * gv.y) + (ov.z * gv.z + ov.w * gv.w);
        }
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);  // D[q] for the lane's column
    const float lse_q = q_ok ? a.lse[qrow * a.H + h] : 0.f;
    if (q_ok && g4 == 0) a.dvec[qrow * a.H + h] = dsum;

    f32x4 dqacc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) dqacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt0 = 0; kt0 < sk_loop; kt0 += 16) {
        const int nrows = min(16, geo.Sk - kt0);
        stage_sync<RAG>();
        if (unit_ok) {
            for (int idx = lane; idx < 16 * F4; idx += 64) {
                const int r = idx / F4, c4 = idx - r * F4;
                float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
                if (r < nrows) {
                    const long long row = krow0 + (long long)(kt0 + r) * geo.k_rs;
                    kv = *reinterpret_cast<const float4*>(a.k + row * a.ldk + h * DH + c4 * 4);
                    vv = *reinterpret_cast<const float4*>(a.v + row * a.ldv + h * DH + c4 * 4);
                }
                *reinterpret_cast<float4*>(&Ks[r * LD + c4 * 4]) = kv;
                *reinterpret_cast<float4*>(&Vs[r * LD + c4 * 4]) = vv;
            }
        }
        stage_sync<RAG>();
        // S^T[key][q] and dP^T[key][q]: A = K / V rows (b128), B = q / dO fragments
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
        const float* kp = &Ks[c16 * LD + 4 * g4];
        const float* vp = &Vs[c16 * LD + 4 * g4];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 kf = *reinterpret_cast<const float4*>(kp + c * 16);
            const float4 vf = *reinterpret_cast<const float4*>(vp + c * 16);
            if (c & 1) {
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[c].x, s1, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, dof[c].x, p1, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[c].y, s1, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, dof[c].y, p1, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[c].z, s1, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, dof[c].z, p1, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[c].w, s1, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, dof[c].w, p1, 0, 0, 0);
            } else {
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[c].x, s0, 0, 0, 0);
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, dof[c].x, p0, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[c].y, s0, 0, 0, 0);
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, dof[c].y, p0, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[c].z, s0, 0, 0, 0);
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, dof[c].z, p0, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[c].w, s0, 0, 0, 0);
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, dof[c].w, p0, 0, 0, 0);
            }
        }
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool kok = kt0 + 4 * g4 + r < geo.Sk;
            const float p = (kok && q_ok) ? __expf((s0[r] + s1[r]) - lse_q) : 0.f;
            float dp = p0[r] + p1[r];
            if (a.drop.enabled)  // O = (P o mask / (1-p)) V: dP = (dO V^T) o mask / (1-p); D = dO . O is unchanged
                dp = dropout_keep(a.drop, ((unsigned long long)(grp * a.H + h) * geo.Sq + qi) * geo.Sk + kt0 + 4 * g4 + r)
                         ? dp * a.drop.scale : 0.f;
            ds[r] = p * (dp - dsum);
        }
        // dQ^T[d][q] += sum_key K[key][d] * dS^T[key][q]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* kr = &Ks[(4 * g4 + r) * LD + c16];
#pragma unroll
            for (int c = 0; c < NC; ++c) dqacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[c * 16], ds[r], dqacc[c], 0, 0, 0);
        }
    }
    if (q_ok) {
        float* dp = a.dq + qrow * a.ld_dq + h * DH + 4 * g4;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            *reinterpret_cast<float4*>(dp + c * 16) = make_float4(dqacc[c][0] * a.scale, dqacc[c][1] * a.scale,
                                                                  dqacc[c][2] * a.scale, dqacc[c][3] * a.scale);
    }
}

template <int DH, bool RAG = false>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnBwdArgs a) {
    constexpr int NC = DH / 16, LD = DH + 4, F4 = DH / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    float* Qs = smem + wave * 2 * 16 * LD;
    float* Gs = Qs + 16 * LD;  // dO rows

    const long long unit = (long long)blockIdx.x * 4 + wave;
    const bool unit_ok = unit < (long long)a.G * a.H * a.ntile;
    const int kt = unit_ok ? (int)(unit % a.ntile) : 0;
    const long long gh = unit_ok ? unit / a.ntile : 0;
    const int h = (int)(gh % a.H), grp = (int)(gh / a.H);
    const BwdGeo geo = bwd_geo(a, grp);
    const long long qrow0 = geo.q0, krow0 = geo.k0;
    const int kj = kt * 16 + c16;
    const bool k_ok = unit_ok && kj < geo.Sk;
    if constexpr (RAG) {
        if (!unit_ok || kt * 16 >= geo.Sk) return;
    }
    const int sq_loop = RAG ? geo.Sq : a.Sq;
    const long long krow = krow0 + (long long)(k_ok ? kj : 0) * geo.k_rs;

    float4 kf[NC], vf[NC];
    {
        const float* kp = a.k + krow * a.ldk + h * DH + 4 * g4;
        const float* vp = a.v + krow * a.ldv + h * DH + 4 * g4;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 kl = *reinterpret_cast<const float4*>(kp + c * 16);  // the row is clamped: load, then zero the VALUE (a select
            const float4 kv = k_ok ? kl : z;                                  // of POINTERS goes through a scratch copy of z + a flat load)
            kf[c] = make_float4(kv.x * a.scale, kv.y * a.scale, kv.z * a.scale, kv.w * a.scale);
            const float4 vl = *reinterpret_cast<const float4*>(vp + c * 16);
            vf[c] = k_ok ? vl : z;
        }
    }
    f32x4 dkacc[NC], dvacc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) dkacc[c] = dvacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int qt0 = 0; qt0 < sq_loop; qt0 += 16) {
        const int nrows = min(16, geo.Sq - qt0);
        stage_sync<RAG>();
        if (unit_ok) {
            for (int idx = lane; idx < 16 * F4; idx += 64) {
                const int r = idx / F4, c4 = idx - r * F4;
                float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv;
                if (r < nrows) {
                    const long long row = qrow0 + (long long)(qt0 + r) * geo.q_rs;
                    qv = *reinterpret_cast<const float4*>(a.q + row * a.ldq + h * DH + c4 * 4);
                    gv = *reinterpret_cast<const float4*>(a.dout + row * a.ldo + h * DH + c4 * 4);
                }
                *reinterpret_cast<float4*>(&Qs[r * LD + c4 * 4]) = qv;
                *reinterpret_cast<float4*>(&Gs[r * LD + c4 * 4]) = gv;
            }
        }
        stage_sync<RAG>();
        // S[q][key] and dP[q][key]: A = Q / dO rows (b128), B = k / v fragments; lane gets q = 4*g4 + r, key = c16
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
        const float* qp = &Qs[c16 * LD + 4 * g4];
        const float* gp = &Gs[c16 * LD + 4 * g4];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 qv = *reinterpret_cast<const float4*>(qp + c * 16);
            const float4 gv = *reinterpret_cast<const float4*>(gp + c * 16);
            if (c & 1) {
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.x, kf[c].x, s1, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.x, vf[c].x, p1, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.y, kf[c].y, s1, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.y, vf[c].y, p1, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.z, kf[c].z, s1, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.z, vf[c].z, p1, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.w, kf[c].w, s1, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.w, vf[c].w, p1, 0, 0, 0);
            } else {
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.x, kf[c].x, s0, 0, 0, 0);
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.x, vf[c].x, p0, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.y, kf[c].y, s0, 0, 0, 0);
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.y, vf[c].y, p0, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.z, kf[c].z, s0, 0, 0, 0);
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.z, vf[c].z, p0, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.w, kf[c].w, s0, 0, 0, 0);
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.w, vf[c].w, p0, 0, 0, 0);
            }
        }
        float pr[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt0 + 4 * g4 + r;
            const bool ok = k_ok && q < geo.Sq;
            float lse = 0.f, dv = 0.f;
            if (ok) {
                const long long row = qrow0 + (long long)q * geo.q_rs;
                lse = a.lse[row * a.H + h];
                dv = a.dvec[row * a.H + h];
            }
            pr[r] = ok ? __expf((s0[r] + s1[r]) - lse) : 0.f;
            float dp = p0[r] + p1[r];
            float keep = 1.f;
            if (a.drop.enabled)
                keep = dropout_keep(a.drop, ((unsigned long long)(grp * a.H + h) * geo.Sq + q) * geo.Sk + kj) ? a.drop.scale : 0.f;
            ds[r] = pr[r] * (dp * keep - dv);
            pr[r] *= keep;  // dV uses the dropped probabilities
        }
        // dV^T[d][key] += sum_q dO[q][d] P[q][key];  dK^T[d][key] += sum_q Q[q][d] dS[q][key]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* gr = &Gs[(4 * g4 + r) * LD + c16];
            const float* qr = &Qs[(4 * g4 + r) * LD + c16];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                dvacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(gr[c * 16], pr[r], dvacc[c], 0, 0, 0);
                dkacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(qr[c * 16], ds[r], dkacc[c], 0, 0, 0);
            }
        }
    }
    if (k_ok) {
        float* dkp = a.dk + krow * a.ld_dk + h * DH + 4 * g4;
        float* dvp = a.dv + krow * a.ld_dv + h * DH + 4 * g4;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            *reinterpret_cast<float4*>(dkp + c * 16) = make_float4(dkacc[c][0] * a.scale, dkacc[c][1] * a.scale,
                                                                   dkacc[c][2] * a.scale, dkacc[c][3] * a.scale);
            *reinterpret_cast<float4*>(dvp + c * 16) = make_float4(dvacc[c][0], dvacc[c][1], dvacc[c][2], dvacc[c][3]);
        }
    }
}

// ---- block-shared staging (round 2) --------------------------------------------------------------------------------------
// The two kernels above give every WAVE its own LDS slice and stage the streamed side (K/V rows for dQ, Q/dO rows for dK/dV)
// per wave, 16 rows at a time, between two block barriers: the four waves of a unit each read the unit's whole K and V (or
// Q and dO) through L2, and no load overlaps a MFMA.  Here a BLOCK owns (unit, 64 queries) for dQ or (unit, 64 keys) for
// dK/dV: the 16-row tile of the streamed side is staged ONCE per block by all 256 threads into one of two LDS buffers, the
// next tile travels in registers while the current one is multiplied (one barrier per tile), the arithmetic is unchanged.
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dq_blk_kernel(const AttnBwdArgs a) {
    constexpr int NC = DH / 16, LD = DH + 4, F4 = DH / 4, PT = 2 * 16 * F4 / 256;  // float4 per thread and tile (K + V)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    const int nqc = (a.Sq + 63) / 64;
    const long long gh = blockIdx.x / nqc;
    const int qc = (int)(blockIdx.x - gh * nqc);
    const int h = (int)(gh % a.H), grp = (int)(gh / a.H);
    const BwdGeo geo = bwd_geo(a, grp);
    if (qc * 64 >= geo.Sq) return;  // ragged: the grid follows the longest unit (the whole block leaves together)
    const long long qrow0 = geo.q0, krow0 = geo.k0;
    const int qi = qc * 64 + wave * 16 + c16;
    const bool q_ok = qi < geo.Sq;
    const long long qrow = qrow0 + (long long)(q_ok ? qi : 0) * geo.q_rs;

    float4 qf[NC], dof[NC];
    float dsum = 0.f;
    {
        const float* qp = a.q + qrow * a.ldq + h * DH + 4 * g4;
        const float* op = a.o + qrow * a.ldo + h * DH + 4 * g4;
        const float* gp = a.dout + qrow * a.ldo + h * DH + 4 * g4;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 qv = q_ok ? *reinterpret_cast<const float4*>(qp + c * 16) : z;
            const float4 ov = q_ok ? *reinterpret_cast<const float4*>(op + c * 16) : z;
            const float4 gv = q_ok ? *reinterpret_cast<const float4*>(gp + c * 16) : z;
            qf[c] = make_float4(qv.x * a.scale, qv.y * a.scale, qv.z * a.scale, qv.w * a.scale);
            dof[c] = gv;
            dsum += (ov.x * gv.x + ov.y * gv.y) + (ov.z * gv.z + ov.w * gv.w);
        }
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);  // D[q] for the lane's column
    const float lse_q = q_ok ? a.lse[qrow * a.H + h] : 0.f;
    if (q_ok && g4 == 0) a.dvec[qrow * a.H + h] = dsum;

    f32x4 dqacc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) dqacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: element e = tid + 256 j of a tile -> (which = K / V, row, float4 column)
    float4 st[PT];
    auto fetch = [&](int kt0) {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int e = tid + 256 * j, which = e / (16 * F4), r = (e % (16 * F4)) / F4, c4 = e % F4;
            st[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kt0 + r < geo.Sk) {
                const long long row = krow0 + (long long)(kt0 + r) * geo.k_rs;
                st[j] = which ? *reinterpret_cast<const float4*>(a.v + row * a.ldv + h * DH + c4 * 4)
                              : *reinterpret_cast<const float4*>(a.k + row * a.ldk + h * DH + c4 * 4);
            }
        }
    };
    auto stash = [&](float* buf) {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int e = tid + 256 * j, which = e / (16 * F4), r = (e % (16 * F4)) / F4, c4 = e % F4;
            *reinterpret_cast<float4*>(&buf[(which * 16 + r) * LD + c4 * 4]) = st[j];
        }
    };
    fetch(0);
    int it = 0;
    for (int kt0 = 0; kt0 < geo.Sk; kt0 += 16, ++it) {
        float* Ks = smem + (it & 1) * 2 * 16 * LD;
        float* Vs = Ks + 16 * LD;
        stash(Ks);
        __syncthreads();
        if (kt0 + 16 < geo.Sk) fetch(kt0 + 16);
        // S^T[key][q] and dP^T[key][q]: A = K / V rows (b128), B = q / dO fragments
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
        const float* kp = &Ks[c16 * LD + 4 * g4];
        const float* vp = &Vs[c16 * LD + 4 * g4];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 kf = *reinterpret_cast<const float4*>(kp + c * 16);
            const float4 vf = *reinterpret_cast<const float4*>(vp + c * 16);
            f32x4& ss = (c & 1) ? s1 : s0;
            f32x4& pp = (c & 1) ? p1 : p0;
            ss = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[c].x, ss, 0, 0, 0);
            pp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, dof[c].x, pp, 0, 0, 0);
            ss = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[c].y, ss, 0, 0, 0);
            pp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, dof[c].y, pp, 0, 0, 0);
            ss = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[c].z, ss, 0, 0, 0);
            pp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, dof[c].z, pp, 0, 0, 0);
            ss = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[c].w, ss, 0, 0, 0);
            pp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, dof[c].w, pp, 0, 0, 0);
        }
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool kok = kt0 + 4 * g4 + r < geo.Sk;
            const float p = (kok && q_ok) ? __expf((s0[r] + s1[r]) - lse_q) : 0.f;
            float dp = p0[r] + p1[r];
            if (a.drop.enabled)  // O = (P o mask / (1-p)) V: dP = (dO V^T) o mask / (1-p); D = dO . O is unchanged
                dp = dropout_keep(a.drop, ((unsigned long long)(grp * a.H + h) * geo.Sq + qi) * geo.Sk + kt0 + 4 * g4 + r)
                         ? dp * a.drop.scale : 0.f;
            ds[r] = p * (dp - dsum);
        }
        // dQ^T[d][q] += sum_key K[key][d] * dS^T[key][q]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* kr = &Ks[(4 * g4 + r) * LD + c16];
#pragma unroll
            for (int c = 0; c < NC; ++c) dqacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[c * 16], ds[r], dqacc[c], 0, 0, 0);
        }
    }
    if (q_ok) {
        float* dp = a.dq + qrow * a.ld_dq + h * DH + 4 * g4;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            *reinterpret_cast<float4*>(dp + c * 16) = make_float4(dqacc[c][0] * a.scale, dqacc[c][1] * a.scale,
                                                                  dqacc[c][2] * a.scale, dqacc[c][3] * a.scale);
    }
}

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_blk_kernel(const AttnBwdArgs a) {
    constexpr int NC = DH / 16, LD = DH + 4, F4 = DH / 4, PT = 2 * 16 * F4 / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    const int nkc = (a.Sk + 63) / 64;
    const long long gh = blockIdx.x / nkc;
    const int kc = (int)(blockIdx.x - gh * nkc);
    const int h = (int)(gh % a.H), grp = (int)(gh / a.H);
    const BwdGeo geo = bwd_geo(a, grp);
    if (kc * 64 >= geo.Sk) return;  // ragged: the grid follows the longest unit
    const long long qrow0 = geo.q0, krow0 = geo.k0;
    const int kj = kc * 64 + wave * 16 + c16;
    const bool k_ok = kj < geo.Sk;
    const long long krow = krow0 + (long long)(k_ok ? kj : 0) * geo.k_rs;

    float4 kf[NC], vf[NC];
    {
        const float* kp = a.k + krow * a.ldk + h * DH + 4 * g4;
        const float* vp = a.v + krow * a.ldv + h * DH + 4 * g4;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 kl = *reinterpret_cast<const float4*>(kp + c * 16);  // the row is clamped: load, then zero the VALUE (a select
            const float4 kv = k_ok ? kl : z;                                  // of POINTERS goes through a scratch copy of z + a flat load)
            kf[c] = make_float4(kv.x * a.scale, kv.y * a.scale, kv.z * a.scale, kv.w * a.scale);
            const float4 vl = *reinterpret_cast<const float4*>(vp + c * 16);
            vf[c] = k_ok ? vl : z;
        }
    }
    f32x4 dkacc[NC], dvacc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) dkacc[c] = dvacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 st[PT];
    auto fetch = [&](int qt0) {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int e = tid + 256 * j, which = e / (16 * F4), r = (e % (16 * F4)) / F4, c4 = e % F4;
            st[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qt0 + r < geo.Sq) {
                const long long row = qrow0 + (long long)(qt0 + r) * geo.q_rs;
                st[j] = which ? *reinterpret_cast<const float4*>(a.dout + row * a.ldo + h * DH + c4 * 4)
                              : *reinterpret_cast<const float4*>(a.q + row * a.ldq + h * DH + c4 * 4);
            }
        }
    };
    auto stash = [&](float* buf) {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int e = tid + 256 * j, which = e / (16 * F4), r = (e % (16 * F4)) / F4, c4 = e % F4;
            *reinterpret_cast<float4*>(&buf[(which * 16 + r) * LD + c4 * 4]) = st[j];
        }
    };
    fetch(0);
    int it = 0;
    for (int qt0 = 0; qt0 < geo.Sq; qt0 += 16, ++it) {
        float* Qs = smem + (it & 1) * 2 * 16 * LD;
        float* Gs = Qs + 16 * LD;  // dO rows
        stash(Qs);
        __syncthreads();
        if (qt0 + 16 < geo.Sq) fetch(qt0 + 16);
        // S[q][key] and dP[q][key]: A = Q / dO rows (b128), B = k / v fragments; lane gets q = 4*g4 + r, key = c16
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
        const float* qp = &Qs[c16 * LD + 4 * g4];
        const float* gp = &Gs[c16 * LD + 4 * g4];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 qv = *reinterpret_cast<const float4*>(qp + c * 16);
            const float4 gv = *reinterpret_cast<const float4*>(gp + c * 16);
            f32x4& ss = (c & 1) ? s1 : s0;
            f32x4& pp = (c & 1) ? p1 : p0;
            ss = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.x, kf[c].x, ss, 0, 0, 0);
            pp = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.x, vf[c].x, pp, 0, 0, 0);
            ss = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.y, kf[c].y, ss, 0, 0, 0);
            pp = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.y, vf[c].y, pp, 0, 0, 0);
            ss = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.z, kf[c].z, ss, 0, 0, 0);
            pp = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.z, vf[c].z, pp, 0, 0, 0);
            ss = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.w, kf[c].w, ss, 0, 0, 0);
            pp = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.w, vf[c].w, pp, 0, 0, 0);
        }
        float pr[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt0 + 4 * g4 + r;
            const bool ok = k_ok && q < geo.Sq;
            float lse = 0.f, dv = 0.f;
            if (ok) {
                const long long row = qrow0 + (long long)q * geo.q_rs;
                lse = a.lse[row * a.H + h];
                dv = a.dvec[row * a.H + h];
            }
            pr[r] = ok ? __expf((s0[r] + s1[r]) - lse) : 0.f;
            const float dp = p0[r] + p1[r];
            float keep = 1.f;
            if (a.drop.enabled)
                keep = dropout_keep(a.drop, ((unsigned long long)(grp * a.H + h) * geo.Sq + q) * geo.Sk + kj) ? a.drop.scale : 0.f;
            ds[r] = pr[r] * (dp * keep - dv);
            pr[r] *= keep;  // dV uses the dropped probabilities
        }
        // dV^T[d][key] += sum_q dO[q][d] P[q][key];  dK^T[d][key] += sum_q Q[q][d] dS[q][key]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* gr = &Gs[(4 * g4 + r) * LD + c16];
            const float* qr = &Qs[(4 * g4 + r) * LD + c16];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                dvacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(gr[c * 16], pr[r], dvacc[c], 0, 0, 0);
                dkacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(qr[c * 16], ds[r], dkacc[c], 0, 0, 0);
            }
        }
    }
    if (k_ok) {
        float* dkp = a.dk + krow * a.ld_dk + h * DH + 4 * g4;
        float* dvp = a.dv + krow * a.ld_dv + h * DH + 4 * g4;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            *reinterpret_cast<float4*>(dkp + c * 16) = make_float4(dkacc[c][0] * a.scale, dkacc[c][1] * a.scale,
                                                                   dkacc[c][2] * a.scale, dkacc[c][3] * a.scale);
            *reinterpret_cast<float4*>(dvp + c * 16) = make_float4(dvacc[c][0], dvacc[c][1], dvacc[c][2], dvacc[c][3]);
        }
    }
}

// ---- sequences of at most 4 steps (motion attention over T' = 4 at the headline shape): ONE pass -------------------------
// A (track, head) unit is 4 rows each of q, k, v, o, dO in and of dQ, dK, dV out - pure streaming.  The two-pass kernels above
// pad it to a 16 x 16 tile, stage it through LDS per wave and read q, k, v, dO twice: 406 us per launch pair at 64 samples
// where the traffic (8 tensors once) takes ~110 us.  Here, as in attn_fwd_small_kernel, half a wave owns a unit: lane c holds
// float4 chunk c of every row, the TT x TT scores S = q k^T, dP = dO v^T and D = dO . o are per-lane partial dot products
// all-reduced over the 32 lanes (DPP, common.h), P is recomputed from the saved log-sum-exp, dS = P o (dP - D), and dQ, dK, dV
// are TT x TT combinations of the rows in registers.  Dropout on P as in the forward (same counter-based mask).
// IO16 (round 6): bf16 q / k / v rows in, bf16 dQ / dK / dV rows out (AttnBwdArgs::io16) - the loads and the stores differ, the arithmetic
// does not (f32 on the widened values: bit-identical to the f32 instantiation on them, gradients rounded once).
template <int TT, bool IO16 = false, bool G16 = false, bool O16 = false>
__global__ __launch_bounds__(256) void attn_bwd_small_kernel(const AttnBwdArgs a) {
    constexpr int DH = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, hw = lane >> 5;
    const long long unit = ((long long)blockIdx.x * 4 + wave) * 2 + hw;
    const long long n_units = (long long)a.G * a.H;
    const bool live = unit < n_units;
    const long long uu = live ? unit : 0;
    const int grp = (int)(uu / a.H), h = (int)(uu - (long long)grp * a.H);
    const BwdGeo geo = bwd_geo(a, grp);
    const long long q0 = geo.q0, k0 = geo.k0;
    const int Sq = live ? geo.Sq : 0, Sk = live ? geo.Sk : 0;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 qv[TT], kv[TT], vv[TT], gv[TT];
    float dsum[TT], lse[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        // rows past the unit's length: a CLAMPED row is loaded and the VALUE zeroed (a `cond ? *p : z` makes the compiler select
        // between the global pointer and a scratch copy of z and load through a flat address - attn_simple.hip)
        const long long qr = q0 + (long long)(t < Sq ? t : 0) * geo.q_rs, kr = k0 + (long long)(t < Sk ? t : 0) * geo.k_rs;
        auto ld = [&](const float* p, bool ok) {
            const float4 v = *reinterpret_cast<const float4*>(p);
            return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        };
        auto ld16 = [&](const float* base, long long off, bool ok) {  // four bfloat16 values of a bf16 matrix
            const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
            return bf16x4_to_f32(make_uint2(ok ? w.x : 0u, ok ? w.y : 0u));
        };
        if constexpr (IO16) {
            qv[t] = ld16(a.q, qr * a.ldq + h * DH + 4 * c, t < Sq);
            kv[t] = ld16(a.k, kr * a.ldk + h * DH + 4 * c, t < Sk);
            vv[t] = ld16(a.v, kr * a.ldv + h * DH + 4 * c, t < Sk);
        } else {
            qv[t] = ld(a.q + qr * a.ldq + h * DH + 4 * c, t < Sq);
            kv[t] = ld(a.k + kr * a.ldk + h * DH + 4 * c, t < Sk);
            vv[t] = ld(a.v + kr * a.ldv + h * DH + 4 * c, t < Sk);
        }
        if constexpr (G16) gv[t] = ld16(a.dout, qr * a.ldo + h * DH + 4 * c, t < Sq);
        else gv[t] = ld(a.dout + qr * a.ldo + h * DH + 4 * c, t < Sq);
        float4 ov;
        if constexpr (O16) ov = ld16(a.o, qr * a.ldo + h * DH + 4 * c, t < Sq);
        else ov = ld(a.o + qr * a.ldo + h * DH + 4 * c, t < Sq);
        lse[t] = t < Sq ? a.lse[qr * a.H + h] : 0.f;
        dsum[t] = half_sum32((ov.x * gv[t].x + ov.y * gv[t].y) + (ov.z * gv[t].z + ov.w * gv[t].w));  // D[i] = dO[i] . O[i]
        if (t < Sq && c == 0) a.dvec[qr * a.H + h] = dsum[t];
    }
    float pd[TT][TT], ds[TT][TT];  // P after dropout (the weights of V in the forward), dS
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const float sp = (qv[i].x * kv[j].x + qv[i].y * kv[j].y) + (qv[i].z * kv[j].z + qv[i].w * kv[j].w);
            const float dp = (gv[i].x * vv[j].x + gv[i].y * vv[j].y) + (gv[i].z * vv[j].z + gv[i].w * vv[j].w);
            const float sc = half_sum32(sp) * a.scale;
            float dpr = half_sum32(dp);
            const bool ok = i < Sq && j < Sk;
            const float p = ok ? __expf(sc - lse[i]) : 0.f;
            float pw = p;
            if (a.drop.enabled) {
                const bool keep = dropout_keep(a.drop, ((unsigned long long)(grp * a.H + h) * geo.Sq + i) * geo.Sk + j);
                pw = keep ? p * a.drop.scale : 0.f;
                dpr = keep ? dpr * a.drop.scale : 0.f;
            }
            pd[i][j] = pw;
            ds[i][j] = p * (dpr - dsum[i]);
        }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        if (i >= Sq) break;
        float4 dq = z;
#pragma unroll
        for (int j = 0; j < TT; ++j) {
            const float w = ds[i][j] * a.scale;
            dq.x += w * kv[j].x; dq.y += w * kv[j].y; dq.z += w * kv[j].z; dq.w += w * kv[j].w;
        }
        const long long eo = (q0 + (long long)i * geo.q_rs) * a.ld_dq + h * DH + 4 * c;
        if constexpr (IO16) *reinterpret_cast<uint2*>(a.dq16 + eo) = f32x4_to_bf16(dq.x, dq.y, dq.z, dq.w);
        else *reinterpret_cast<float4*>(a.dq + eo) = dq;
    }
#pragma unroll
    for (int j = 0; j < TT; ++j) {
        if (j >= Sk) break;
        float4 dk = z, dv = z;
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            const float w = ds[i][j] * a.scale, pw = pd[i][j];
            dk.x += w * qv[i].x; dk.y += w * qv[i].y; dk.z += w * qv[i].z; dk.w += w * qv[i].w;
            dv.x += pw * gv[i].x; dv.y += pw * gv[i].y; dv.z += pw * gv[i].z; dv.w += pw * gv[i].w;
        }
        const long long kr = k0 + (long long)j * geo.k_rs;
        if constexpr (IO16) {  // (a.dk / a.dv hold the bf16 matrices: launch_attention_bwd)
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.dk) + kr * a.ld_dk + h * DH + 4 * c) = f32x4_to_bf16(dk.x, dk.y, dk.z, dk.w);
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.dv) + kr * a.ld_dv + h * DH + 4 * c) = f32x4_to_bf16(dv.x, dv.y, dv.z, dv.w);
        } else {
            *reinterpret_cast<float4*>(a.dk + kr * a.ld_dk + h * DH + 4 * c) = dk;
            *reinterpret_cast<float4*>(a.dv + kr * a.ld_dv + h * DH + 4 * c) = dv;
        }
    }
}

// ---- ONE pass for units of at most 128 queries and 128 keys (round 3) ------------------------------------------------------
// The two-pass kernels recompute S and dP in both passes (7 tile products instead of 5) and read q, k, v, dO twice; on a ragged
// batch of short units (inter-object attention over 8..80 tracks, motion attention over 3..25 steps) every wave also stages its
// own copy of the streamed side with no load in flight under its MFMAs: 5.4 ms of a 28 ms step where the traffic takes 1.0 ms and
// the matrix pipe 0.9 ms.  Here a BLOCK owns one (unit, head):
//   * wave w keeps ONE 16-key tile in registers - k / v rows as MFMA operands (kf, vf), K again with the key on the contraction
//     slots (kT) - and its dK / dV accumulators; units of more than 16 * NWU keys walk their key tiles in groups of NWU;
//   * the block walks the unit's 16-query tiles: q and dO rows are staged ONCE per tile for all waves (NWU = 4: the next tile
//     travels in registers under the current tile's MFMAs, two LDS buffers);
//   * per tile a wave computes S = Q K^T and dP = dO V^T for its keys, P from the saved log-sum-exp, dS = P o (dP - D), then
//     dV += P^T dO, dK += dS^T Q, and - dS transposed through 1 KB of wave-private LDS - its partial dQ = dS K;
//   * the NWU partial dQ tiles are summed through LDS in wave order (bit-repeatable: no atomics) and written once; a later key
//     group adds to what the same thread wrote before.
// D = dO . O and the log-sum-exp of the unit's queries sit in LDS.  Same arithmetic, masks and dropout counters as the two-pass
// kernels.  q, k, v, o, dO are read once and dQ, dK, dV written once per (unit, head).
// IO16 (round 6): bf16 q / k / v in, bf16 dQ / dK / dV out (AttnBwdArgs::io16).  Only the loads and the stores differ: q rows are widened on
// their way into the f32 LDS tile (four-wave shape: the next tile's q piece travels in four registers beside the DMA of dO and O), k / v
// fragments are 8-byte loads, the gradients leave as 8-byte rows of four values.
// MF (round 6, with IO16; sola_tune "attn_bwd_bf16_mfma"): the five products run on v_mfma_f32_16x16x16_bf16 - every group of four f32
// MFMAs over the register slots (.x .y .z .w of a row chunk, or r = 0..3 of a C-layout tile) is ONE bf16 MFMA whose operand is those four
// values packed: q / k / v are bfloat16 already (exact), dO, the probabilities and dS are rounded to bfloat16 (nearest even) - the operand
// precision torch's autocast backward has; accumulation, the softmax terms, D = dO . O and the output sums stay f32.  K is NOT pre-scaled
// here (scale * k is not a bfloat16): S and dQ take the scale behind their products.  160 -> 40 matrix instructions per (query tile, key
// tile) at 8x the rate: the f32 matrix time was a third of the kernel on ragged batches.
// G16 (with MF): dO arrives as bfloat16 rows too (AttnBwdArgs::g16) - staged like q: widened into the f32 tile image (register shapes: on the
// way in; four-wave shape: raw DMA, each thread widens its own piece and sums D = dO . O for its row with the fifteen lanes beside it).
// O16 (with G16): O is a bfloat16 matrix as well (AttnBwdArgs::o16): it only enters D = dO . O - register shapes widen it on the way in, the
// four-wave shape fetches it as a raw DMA beside dO's and each thread multiplies its own eight values of the two.
template <int NWU, bool IO16 = false, bool MF = false, bool G16 = false, bool O16 = false>
__global__ __launch_bounds__(64 * NWU, 2) void attn_bwd_fused_kernel(const AttnBwdArgs a) {
    static_assert(!MF || IO16, "the bf16 products take bf16 q / k / v");
    static_assert(!G16 || MF, "bf16 dO goes with the bf16 products");
    static_assert(!O16 || G16, "bf16 O goes with bf16 dO");
    typedef short short4m __attribute__((ext_vector_type(4)));
    auto pk4 = [](float x, float y, float z, float w) -> short4m { return __builtin_bit_cast(short4m, f32x4_to_bf16(x, y, z, w)); };
    auto mf16 = [](const short4m x, const short4m y, const f32x4 c) -> f32x4 { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, c, 0, 0, 0); };
    constexpr int DH = 128, NC = DH / 16, LD = DH + 4, F4 = DH / 4, NT = 64 * NWU;
    // NWU = 4: the next tile's q, dO and O rows travel straight into LDS (global_load_lds, no registers) under the current tile's
    // MFMAs - two tile buffers.  The LDS side of the instruction is lane-linear (base + 16 * lane = two 512-byte rows back to back),
    // so a tile is eight row PAIRS of 1024 + 32 bytes: the b32 column reads stay conflict-free, the b128 row reads see 2-way
    // conflicts (16 reads per tile).  D = dO . O is then summed from LDS.
    // NWU = 1, 2: up to eight / four blocks per CU overlap each other; loads through registers, one buffer, rows padded to LD.
    constexpr bool DMA = NWU == 4;
    constexpr int NBUF = DMA ? 2 : 1, RP = LD;  // row pitch of a staged tile, floats
    constexpr int PER = 16 * F4 / NT;  // float4 per thread per tensor per tile
    constexpr int STAGE_UNROLL = NWU == 2 ? 1 : 2;  // the two-wave shape keeps its staging loop rolled: it spills otherwise
    constexpr int SLOTS = NWU > 1 ? NWU * 16 * LD : 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    float* const Qs = smem;                         // [NBUF][16][RP]
    float* const Gs = Qs + NBUF * 16 * RP;          // [NBUF][16][RP]  dO rows
    float* const Os = Gs + NBUF * 16 * RP;          // [16][RP]        O rows of the tile in flight (DMA only)
    float* const slots = Os + (DMA ? 16 * RP : 0);  // [NWU][16][LD]   partial dQ tiles
    float* const tsc = slots + SLOTS + wave * (16 * 17);
    float* const dsh = slots + SLOTS + NWU * (16 * 17);  // [NBUF][16] D = dO . O of the staged tile's queries
    float* const lsh = dsh + NBUF * 16;                   // [NBUF][16] their log-sum-exp

    const int h = (int)(blockIdx.x % a.H), grp = (int)(blockIdx.x / a.H);
    const BwdGeo geo = bwd_geo(a, grp);
    const int nkt = (geo.Sk + 15) >> 4, nqt_all = (geo.Sq + 15) >> 4;
    const int qt_begin = a.qc_tiles ? (int)blockIdx.y * a.qc_tiles : 0;
    const int nqt = a.qc_tiles ? min(nqt_all, qt_begin + a.qc_tiles) : nqt_all;  // this block's tiles: [qt_begin, nqt)
    if (qt_begin >= nqt) return;  // a chunk past the unit's queries (ragged batches size the grid for the longest unit)

    // element (row r, 16-byte chunk ch) of a staged tile
    auto at = [&](float* base, int buf, int r, int ch) -> float* {
        return DMA ? base + buf * 16 * RP + (r >> 1) * (2 * RP) + (r & 1) * DH + (ch << 2) : base + (buf * 16 + r) * RP + (ch << 2);
    };
    // registers path: a tile's q and dO rows -> LDS; a row is the 32 float4 of half a wave, which also sums D = dO . O for it.
    // SB row pieces per batch: their loads (q, dO, O and the row's log-sum-exp - every lane of the row fetches it, one address) go out back to
    // back, then the batch is summed and stored.  MF shapes have the registers for whole batches of four (the two-wave shape's tile in ONE
    // round trip); the f32-product shapes keep one / two pieces per trip (they spill otherwise).  Round 6: the rolled loop of the two-wave
    // shape was 4 pieces x 2 dependent round trips (the log-sum-exp load sat in a branch of its own behind the others) per 16-query tile.
    constexpr int SBW = MF ? (NWU == 2 ? 2 : 4) : STAGE_UNROLL;  // (four pieces in the two-wave MF shape: 256 registers and four spills)
    constexpr int SB = SBW < PER ? SBW : PER;
    static_assert(PER % SB == 0, "whole batches");
    auto stage_direct = [&](int qt0) {
#pragma unroll 1
        for (int j0 = 0; j0 < PER; j0 += SB) {
            float4 qf[IO16 ? 1 : SB], gv[SB], ov[SB];
            uint2 qr[IO16 ? SB : 1];  // bf16 rows stay raw until they are stored (two registers a piece instead of four)
            float ls[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = tid + (j0 + u) * NT, r = idx / F4, c4 = idx - r * F4;
                const int q = qt0 + r;
                const long long row = geo.q0 + (long long)(q < geo.Sq ? q : 0) * geo.q_rs;
                if constexpr (IO16) qr[u] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.q) + row * a.ldq + h * DH + c4 * 4);
                else qf[u] = *reinterpret_cast<const float4*>(a.q + row * a.ldq + h * DH + c4 * 4);
                if constexpr (G16) gv[u] = bf16x4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.dout) + row * a.ldo + h * DH + c4 * 4));
                else gv[u] = *reinterpret_cast<const float4*>(a.dout + row * a.ldo + h * DH + c4 * 4);
                if constexpr (O16) ov[u] = bf16x4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.o) + row * a.ldo + h * DH + c4 * 4));
                else ov[u] = *reinterpret_cast<const float4*>(a.o + row * a.ldo + h * DH + c4 * 4);
                ls[u] = a.lse[row * a.H + h];
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = tid + (j0 + u) * NT, r = idx / F4, c4 = idx - r * F4;
                const bool in = qt0 + r < geo.Sq;
                float4 qv;
                if constexpr (IO16) qv = bf16x4_to_f32(in ? qr[u] : make_uint2(0u, 0u));
                else qv = in ? qf[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (!in) gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                const float d = half_sum32((ov[u].x * gv[u].x + ov[u].y * gv[u].y) + (ov[u].z * gv[u].z + ov[u].w * gv[u].w));
                if (c4 == 0) {
                    dsh[r] = d;
                    lsh[r] = ls[u];
                }
                *reinterpret_cast<float4*>(at(Qs, 0, r, c4)) = qv;
                *reinterpret_cast<float4*>(at(Gs, 0, r, c4)) = gv[u];
            }
        }
    };
    // DMA path: wave w moves row pairs 2w and 2w+1 of each of the three tensors, one pair (1 KB) per instruction; rows past the
    // unit's queries re-read its first row (finite values; their probabilities are masked to zero).  (Two half-wave instructions
    // per pair under `if (lane < 32) ... else ...` do NOT work: the compiler merges the branches and reads the LDS base of the first
    // lane for the whole wave.)
    // IO16, four waves: the next tile's q rows arrive by DMA as RAW bfloat16 rows at the start of the tile buffer's q image (16 x 256 bytes:
    // wave w moves rows 4w..4w+3, lane l the eight values at 8 * (l & 15) of row 4w + (l >> 4) - lane-linear, so every thread's piece lands
    // at 16 * tid); behind the wait each thread reads ITS OWN piece back (no barrier needed for that), and once every thread has (the
    // barrier that follows anyway) widens it into the f32 image over the raw bytes.  No register lives across the tile's MFMAs.
    auto issue_tile = [&](int qt0, int buf) {
        if constexpr (IO16) {
            const int q = qt0 + (tid >> 4);
            const long long row = geo.q0 + (long long)(q < geo.Sq ? q : 0) * geo.q_rs;
            __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const unsigned short*>(a.q) + row * a.ldq + h * DH + (tid & 15) * 8),
                                             (lptr_t)(Qs + buf * 16 * RP + wave * 256), 16, 0, 0);
            if constexpr (G16)
                __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const unsigned short*>(a.dout) + row * a.ldo + h * DH + (tid & 15) * 8),
                                                 (lptr_t)(Gs + buf * 16 * RP + wave * 256), 16, 0, 0);
            if constexpr (O16)
                __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const unsigned short*>(a.o) + row * a.ldo + h * DH + (tid & 15) * 8),
                                                 (lptr_t)(Os + wave * 256), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pr2 = wave * 2 + i, r = pr2 * 2 + (lane >> 5), ch = lane & 31;
            const int q = qt0 + r;
            const long long row = geo.q0 + (long long)(q < geo.Sq ? q : 0) * geo.q_rs;
            const int pc = buf * 16 * RP + pr2 * (2 * RP);  // floats
            if constexpr (!IO16) __builtin_amdgcn_global_load_lds((gptr_t)(a.q + row * a.ldq + h * DH + ch * 4), (lptr_t)(Qs + pc), 16, 0, 0);
            if constexpr (!G16) __builtin_amdgcn_global_load_lds((gptr_t)(a.dout + row * a.ldo + h * DH + ch * 4), (lptr_t)(Gs + pc), 16, 0, 0);
            if constexpr (!O16) __builtin_amdgcn_global_load_lds((gptr_t)(a.o + row * a.ldo + h * DH + ch * 4), (lptr_t)(Os + pr2 * (2 * RP)), 16, 0, 0);
        }
        if (tid < 16) {  // the rows' log-sum-exp: 4 bytes per lane, straight into the tile's slots
            const int q = qt0 + tid;
            const long long row = geo.q0 + (long long)(q < geo.Sq ? q : 0) * geo.q_rs;
            __builtin_amdgcn_global_load_lds((gptr_t)(a.lse + row * a.H + h), (lptr_t)(lsh + buf * 16), 4, 0, 0);
        }
    };
    auto load_qraw = [&](int buf) -> uint4 { return *reinterpret_cast<const uint4*>(Qs + buf * 16 * RP + tid * 4); };
    auto store_q = [&](int buf, const uint4 w) {
        const int r = tid >> 4, c8 = tid & 15;
        *reinterpret_cast<float4*>(at(Qs, buf, r, 2 * c8)) = bf16x4_to_f32(make_uint2(w.x, w.y));
        *reinterpret_cast<float4*>(at(Qs, buf, r, 2 * c8 + 1)) = bf16x4_to_f32(make_uint2(w.z, w.w));
    };
    auto load_graw = [&](int buf) -> uint4 { return *reinterpret_cast<const uint4*>(Gs + buf * 16 * RP + tid * 4); };
    // G16: this thread's eight dO values of row tid >> 4 into the f32 image, and the row's D = dO . O from them and the O rows the DMA landed
    // (sixteen lanes a row: the sum closes with four exchanges inside the 16-lane group)
    auto store_g = [&](int buf, const uint4 w) {
        const int r = tid >> 4, c8 = tid & 15;
        const float4 g0 = bf16x4_to_f32(make_uint2(w.x, w.y)), g1 = bf16x4_to_f32(make_uint2(w.z, w.w));
        *reinterpret_cast<float4*>(at(Gs, buf, r, 2 * c8)) = g0;
        *reinterpret_cast<float4*>(at(Gs, buf, r, 2 * c8 + 1)) = g1;
        float4 o0, o1;
        if constexpr (O16) {  // this thread's own raw piece of the O rows (same lane-linear place as its dO piece)
            const uint4 wo = *reinterpret_cast<const uint4*>(Os + tid * 4);
            o0 = bf16x4_to_f32(make_uint2(wo.x, wo.y)); o1 = bf16x4_to_f32(make_uint2(wo.z, wo.w));
        } else {
            o0 = *reinterpret_cast<const float4*>(at(Os, 0, r, 2 * c8)); o1 = *reinterpret_cast<const float4*>(at(Os, 0, r, 2 * c8 + 1));
        }
        float d = ((o0.x * g0.x + o0.y * g0.y) + (o0.z * g0.z + o0.w * g0.w)) + ((o1.x * g1.x + o1.y * g1.y) + (o1.z * g1.z + o1.w * g1.w));
        d += __shfl_xor(d, 8, 16);
        d += __shfl_xor(d, 4, 16);
        d += __shfl_xor(d, 2, 16);
        d += __shfl_xor(d, 1, 16);
        if (c8 == 0) dsh[buf * 16 + r] = d;
    };
    auto dvec_from_lds = [&](int buf) {  // D of a landed tile: half a wave per row, two rows per wave and pass
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (wave * 2 + j) * 2 + (lane >> 5), c4 = lane & 31;
            const float4 gv = *reinterpret_cast<const float4*>(at(Gs, buf, r, c4));
            const float4 ov = *reinterpret_cast<const float4*>(at(Os, 0, r, c4));
            const float d = half_sum32((ov.x * gv.x + ov.y * gv.y) + (ov.z * gv.z + ov.w * gv.w));
            if (c4 == 0) dsh[buf * 16 + r] = d;
        }
    };

    for (int kg = 0; kg < nkt; kg += NWU) {
        const int kt = kg + wave;
        const bool w_ok = kt < nkt;
        const int kj = kt * 16 + c16;
        const bool k_ok = w_ok && kj < geo.Sk;
        const long long krow = geo.k0 + (long long)(k_ok ? kj : 0) * geo.k_rs;
        if constexpr (DMA) issue_tile(qt_begin * 16, 0);  // under the K / V loads
        float4 kf[NC], vf[NC];
        float kT[NC][4];
        short4m kfh[NC], vfh[NC], kTh[NC];  // MF: the same three fragment sets as bfloat16 operands - the raw values, no widening
        if constexpr (MF) {
            const unsigned short* kp16 = reinterpret_cast<const unsigned short*>(a.k) + krow * a.ldk + h * DH + 4 * g4;
            const unsigned short* vp16 = reinterpret_cast<const unsigned short*>(a.v) + krow * a.ldv + h * DH + 4 * g4;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const uint2 kl = *reinterpret_cast<const uint2*>(kp16 + c * 16), vl = *reinterpret_cast<const uint2*>(vp16 + c * 16);
                kfh[c] = __builtin_bit_cast(short4m, make_uint2(k_ok ? kl.x : 0u, k_ok ? kl.y : 0u));
                vfh[c] = __builtin_bit_cast(short4m, make_uint2(k_ok ? vl.x : 0u, k_ok ? vl.y : 0u));
            }
            unsigned short kr[NC][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = kt * 16 + 4 * g4 + r;
                const bool okr = w_ok && kk < geo.Sk;
                const long long ro = (geo.k0 + (long long)(okr ? kk : 0) * geo.k_rs) * a.ldk + h * DH + c16;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const unsigned short v = reinterpret_cast<const unsigned short*>(a.k)[ro + c * 16];
                    kr[c][r] = okr ? v : (unsigned short)0;
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
                kTh[c] = __builtin_bit_cast(short4m, make_uint2((unsigned)kr[c][0] | ((unsigned)kr[c][1] << 16), (unsigned)kr[c][2] | ((unsigned)kr[c][3] << 16)));
        } else {
            const float* kp = a.k + krow * a.ldk + h * DH + 4 * g4;
            const float* vp = a.v + krow * a.ldv + h * DH + 4 * g4;
            const unsigned short* kp16 = reinterpret_cast<const unsigned short*>(a.k) + krow * a.ldk + h * DH + 4 * g4;
            const unsigned short* vp16 = reinterpret_cast<const unsigned short*>(a.v) + krow * a.ldv + h * DH + 4 * g4;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float4 kl, vl;  // clamped row, VALUE select (see attn_bwd_dkv_kernel)
                if constexpr (IO16) {
                    kl = bf16x4_to_f32(*reinterpret_cast<const uint2*>(kp16 + c * 16));
                    vl = bf16x4_to_f32(*reinterpret_cast<const uint2*>(vp16 + c * 16));
                } else {
                    kl = *reinterpret_cast<const float4*>(kp + c * 16);
                    vl = *reinterpret_cast<const float4*>(vp + c * 16);
                }
                const float4 kv = k_ok ? kl : z;
                kf[c] = make_float4(kv.x * a.scale, kv.y * a.scale, kv.z * a.scale, kv.w * a.scale);
                vf[c] = k_ok ? vl : z;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // the same tile with the key on the contraction slots: K[key = 4*g4 + r][d = 16*c + c16]
                const int kk = kt * 16 + 4 * g4 + r;
                const bool okr = w_ok && kk < geo.Sk;
                const long long ro = (geo.k0 + (long long)(okr ? kk : 0) * geo.k_rs) * a.ldk + h * DH + c16;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float v;
                    if constexpr (IO16) v = __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(a.k)[ro + c * 16] << 16);
                    else v = a.k[ro + c * 16];
                    kT[c][r] = okr ? v * a.scale : 0.f;
                }
            }
        }
        f32x4 dkacc[NC], dvacc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) dkacc[c] = dvacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (IO16) {
                const uint4 w = load_qraw(0);
                uint4 wg = make_uint4(0u, 0u, 0u, 0u);
                if constexpr (G16) wg = load_graw(0);
                __syncthreads();  // every thread holds its raw piece(s); the O rows are visible
                store_q(0, w);
                if constexpr (G16) store_g(0, wg);
            }
            __syncthreads();  // tile 0 landed; also: the previous group's last reduction done
            if constexpr (!G16) dvec_from_lds(0);
            __syncthreads();
        } else {
            stage_direct(qt_begin * 16);
            __syncthreads();  // also: the previous group's last reduction done
        }
        for (int it = qt_begin; it < nqt; ++it) {
            const int qt0 = it * 16, cur = DMA ? ((it - qt_begin) & 1) : 0;
            const bool more = it + 1 < nqt;
            if constexpr (DMA) {
                if (more && !(a.ntile & 2)) issue_tile(qt0 + 16, cur ^ 1);
            }
            f32x4 dqacc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) dqacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (w_ok && !(a.ntile & 1)) {
                // S[q][key] and dP[q][key]: A = Q / dO rows (b128), B = k / v fragments; lane gets q = 4*g4 + r, key = c16
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float4 qv = *reinterpret_cast<const float4*>(at(Qs, cur, c16, c * 4 + g4));
                    const float4 gv = *reinterpret_cast<const float4*>(at(Gs, cur, c16, c * 4 + g4));
                    if constexpr (MF) {
                        const short4m qh = pk4(qv.x, qv.y, qv.z, qv.w), gh = pk4(gv.x, gv.y, gv.z, gv.w);
                        if (c & 1) { s1 = mf16(qh, kfh[c], s1); p1 = mf16(gh, vfh[c], p1); }
                        else { s0 = mf16(qh, kfh[c], s0); p0 = mf16(gh, vfh[c], p0); }
                    } else if (c & 1) {
                        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.x, kf[c].x, s1, 0, 0, 0);
                        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.x, vf[c].x, p1, 0, 0, 0);
                        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.y, kf[c].y, s1, 0, 0, 0);
                        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.y, vf[c].y, p1, 0, 0, 0);
                        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.z, kf[c].z, s1, 0, 0, 0);
                        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.z, vf[c].z, p1, 0, 0, 0);
                        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.w, kf[c].w, s1, 0, 0, 0);
                        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.w, vf[c].w, p1, 0, 0, 0);
                    } else {
                        s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.x, kf[c].x, s0, 0, 0, 0);
                        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.x, vf[c].x, p0, 0, 0, 0);
                        s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.y, kf[c].y, s0, 0, 0, 0);
                        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.y, vf[c].y, p0, 0, 0, 0);
                        s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.z, kf[c].z, s0, 0, 0, 0);
                        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.z, vf[c].z, p0, 0, 0, 0);
                        s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.w, kf[c].w, s0, 0, 0, 0);
                        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.w, vf[c].w, p0, 0, 0, 0);
                    }
                }
                const float4 lse4 = *reinterpret_cast<const float4*>(&lsh[cur * 16 + 4 * g4]);
                const float4 dv4 = *reinterpret_cast<const float4*>(&dsh[cur * 16 + 4 * g4]);
                const float lse_r[4] = {lse4.x, lse4.y, lse4.z, lse4.w}, dv_r[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
                float pr[4], ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qt0 + 4 * g4 + r;
                    const bool ok = k_ok && q < geo.Sq;
                    pr[r] = ok ? __expf((s0[r] + s1[r]) * (MF ? a.scale : 1.f) - lse_r[r]) : 0.f;
                    const float dp = p0[r] + p1[r];
                    float keep = 1.f;
                    if (a.drop.enabled)
                        keep = dropout_keep(a.drop, ((unsigned long long)(grp * a.H + h) * geo.Sq + q) * geo.Sk + kj) ? a.drop.scale : 0.f;
                    ds[r] = ok ? pr[r] * (dp * keep - dv_r[r]) : 0.f;
                    pr[r] *= keep;  // dV uses the dropped probabilities
                }
                // dV^T[d][key] += sum_q dO[q][d] P[q][key];  dK^T[d][key] += sum_q Q[q][d] dS[q][key]
                if constexpr (MF) {
                    const short4m prh = pk4(pr[0], pr[1], pr[2], pr[3]), dsh4 = pk4(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int ch = c * 4 + (c16 >> 2), e = c16 & 3, R0 = 4 * g4;
                        const float g0 = at(Gs, cur, R0, ch)[e], g1 = at(Gs, cur, R0 + 1, ch)[e], g2 = at(Gs, cur, R0 + 2, ch)[e], g3 = at(Gs, cur, R0 + 3, ch)[e];
                        const float q0 = at(Qs, cur, R0, ch)[e], q1 = at(Qs, cur, R0 + 1, ch)[e], q2 = at(Qs, cur, R0 + 2, ch)[e], q3 = at(Qs, cur, R0 + 3, ch)[e];
                        dvacc[c] = mf16(pk4(g0, g1, g2, g3), prh, dvacc[c]);
                        dkacc[c] = mf16(pk4(q0, q1, q2, q3), dsh4, dkacc[c]);
                    }
                } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int R = 4 * g4 + r;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const float gcol = at(Gs, cur, R, c * 4 + (c16 >> 2))[c16 & 3];
                        const float qcol = at(Qs, cur, R, c * 4 + (c16 >> 2))[c16 & 3];
                        dvacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(gcol, pr[r], dvacc[c], 0, 0, 0);
                        dkacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(qcol, ds[r], dkacc[c], 0, 0, 0);
                    }
                }
                }
                // dS with the KEY on the contraction slots: through the wave's own 16 x 17 words of LDS
#pragma unroll
                for (int r = 0; r < 4; ++r) tsc[(4 * g4 + r) * 17 + c16] = ds[r];
                stage_sync<true>();
                float dst[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r] = tsc[c16 * 17 + 4 * g4 + r];  // dS[q = c16][key = 4*g4 + r]
                stage_sync<true>();  // the next tile's writes stay behind these reads
                // dQ^T[d][q] += sum_key K[key][d] * dS[q][key]   (K pre-scaled)
                if constexpr (MF) {
                    const short4m dsth = pk4(dst[0], dst[1], dst[2], dst[3]);
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        dqacc[c] = mf16(kTh[c], dsth, dqacc[c]);
                        dqacc[c] *= a.scale;  // K was not pre-scaled
                    }
                } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < NC; ++c) dqacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kT[c][r], dst[r], dqacc[c], 0, 0, 0);
                }
                if constexpr (NWU > 1) {
                    float* sl = slots + wave * 16 * LD + c16 * LD + 4 * g4;
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        *reinterpret_cast<float4*>(sl + c * 16) = make_float4(dqacc[c][0], dqacc[c][1], dqacc[c][2], dqacc[c][3]);
                }
            }
            if constexpr (NWU > 1) {
                if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile has landed
                uint4 qraw = make_uint4(0u, 0u, 0u, 0u), graw = qraw;
                if constexpr (DMA && IO16) {
                    if (more && !(a.ntile & 2)) {
                        qraw = load_qraw(cur ^ 1);  // this thread's own DMA piece
                        if constexpr (G16) graw = load_graw(cur ^ 1);
                    }
                }
                __syncthreads();  // partial tiles complete (DMA: and the next tile visible; IO16: every raw q piece is in a register)
                if constexpr (DMA && IO16) {
                    if (more && !(a.ntile & 2)) {
                        store_q(cur ^ 1, qraw);  // nobody reads that buffer's q image before the iteration's closing barrier
                        if constexpr (G16) store_g(cur ^ 1, graw);  // ... nor its dO image / D slots
                    }
                }
                const int nact = min(NWU, nkt - kg);
#pragma unroll 1
                for (int j = 0; j < PER; ++j) {
                    const int idx = tid + j * NT, r = idx / F4, c4 = idx - r * F4;
                    float4 acc = *reinterpret_cast<const float4*>(&slots[r * LD + c4 * 4]);
                    for (int w = 1; w < nact; ++w) {
                        const float4 v = *reinterpret_cast<const float4*>(&slots[(w * 16 + r) * LD + c4 * 4]);
                        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                    }
                    const int q = qt0 + r;
                    if (q < geo.Sq) {
                        const long long eo = (geo.q0 + (long long)q * geo.q_rs) * a.ld_dq + h * DH + c4 * 4;
                        float* p = a.dq + eo;
                        if (kg > 0) {  // written by this same thread in the previous key group
                            const float4 old = *reinterpret_cast<const float4*>(p);
                            acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
                        }
                        if constexpr (IO16) {  // the f32 row is scratch between key groups; the last group writes the bf16 row
                            if (kg + NWU < nkt) *reinterpret_cast<float4*>(p) = acc;
                            else *reinterpret_cast<uint2*>(a.dq16 + eo) = f32x4_to_bf16(acc.x, acc.y, acc.z, acc.w);
                        } else {
                            *reinterpret_cast<float4*>(p) = acc;
                        }
                    }
                }
                if constexpr (DMA) {
                    if constexpr (!G16) {
                        if (more && !(a.ntile & 2)) dvec_from_lds(cur ^ 1);
                    }
                } else {
                    if (more && !(a.ntile & 2)) stage_direct(qt0 + 16);
                }
                __syncthreads();  // the partial tiles are free again, the next tile's D / lse (registers path: the tile itself) in place
            } else {
                const int q = qt0 + c16;
                if (q < geo.Sq) {  // one wave, one key tile: dqacc is the whole dQ tile
                    const long long eo = (geo.q0 + (long long)q * geo.q_rs) * a.ld_dq + h * DH + 4 * g4;
                    float* p = a.dq + eo;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        float4 acc = make_float4(dqacc[c][0], dqacc[c][1], dqacc[c][2], dqacc[c][3]);
                        if (kg > 0) {
                            const float4 old = *reinterpret_cast<const float4*>(p + c * 16);
                            acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
                        }
                        if constexpr (IO16) {
                            if (kg + NWU < nkt) *reinterpret_cast<float4*>(p + c * 16) = acc;
                            else *reinterpret_cast<uint2*>(a.dq16 + eo + c * 16) = f32x4_to_bf16(acc.x, acc.y, acc.z, acc.w);
                        } else {
                            *reinterpret_cast<float4*>(p + c * 16) = acc;
                        }
                    }
                }
                __syncthreads();  // every lane is done with the tile
                if (more) stage_direct(qt0 + 16);
                __syncthreads();
            }
        }
        if (k_ok) {
            float* dkp = a.dk + krow * a.ld_dk + h * DH + 4 * g4;
            float* dvp = a.dv + krow * a.ld_dv + h * DH + 4 * g4;
            if (a.qc_tiles) {  // this chunk's partial sums (one key group: the launcher chunks units of at most 16 * NWU keys only)
                const long long slot = geo.q0 / (16 * a.qc_tiles) + grp + blockIdx.y;
                dkp = a.part + ((slot * a.H + h) * 2 * (16 * NWU) + kj) * DH + 4 * g4;
                dvp = dkp + (16 * NWU) * DH;
            }
            if (IO16 && !a.qc_tiles) {  // final rows: bf16
                unsigned short* dk16 = reinterpret_cast<unsigned short*>(a.dk) + krow * a.ld_dk + h * DH + 4 * g4;
                unsigned short* dv16 = reinterpret_cast<unsigned short*>(a.dv) + krow * a.ld_dv + h * DH + 4 * g4;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    *reinterpret_cast<uint2*>(dk16 + c * 16) = f32x4_to_bf16(dkacc[c][0] * a.scale, dkacc[c][1] * a.scale, dkacc[c][2] * a.scale, dkacc[c][3] * a.scale);
                    *reinterpret_cast<uint2*>(dv16 + c * 16) = f32x4_to_bf16(dvacc[c][0], dvacc[c][1], dvacc[c][2], dvacc[c][3]);
                }
            } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                *reinterpret_cast<float4*>(dkp + c * 16) = make_float4(dkacc[c][0] * a.scale, dkacc[c][1] * a.scale,
                                                                       dkacc[c][2] * a.scale, dkacc[c][3] * a.scale);
                *reinterpret_cast<float4*>(dvp + c * 16) = make_float4(dvacc[c][0], dvacc[c][1], dvacc[c][2], dvacc[c][3]);
            }
            }
        }
    }
}

// dK / dV of a chunked launch: the partial sums of a unit's chunks, added in chunk order
__global__ __launch_bounds__(256) void attn_bwd_part_reduce_kernel(const AttnBwdArgs a, int kp) {
    constexpr int DH = 128;
    const int h = (int)(blockIdx.x % a.H), grp = (int)(blockIdx.x / a.H);
    const BwdGeo geo = bwd_geo(a, grp);
    const int nch = ((geo.Sq + 15) / 16 + a.qc_tiles - 1) / a.qc_tiles;
    const long long slot0 = geo.q0 / (16 * a.qc_tiles) + grp;
    // blockIdx.y: 256 of the unit's Sk * 64 float4 each (one sample per step has eight (unit, head) pairs: eight blocks walked twelve
    // positions x four chunks one load at a time, 16 us)
    for (int idx = blockIdx.y * 256 + threadIdx.x; idx < geo.Sk * 2 * (DH / 4); idx += 256 * gridDim.y) {
        const int c4 = idx & 31, t = (idx >> 5) & 1, kj = idx >> 6;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int c = 0; c < nch; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(a.part + (((slot0 + c) * a.H + h) * 2 + t) * (long long)kp * DH + kj * DH + c4 * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const long long krow = geo.k0 + (long long)kj * geo.k_rs;
        if (a.io16) {
            unsigned short* p16 = (t ? reinterpret_cast<unsigned short*>(a.dv) + krow * a.ld_dv : reinterpret_cast<unsigned short*>(a.dk) + krow * a.ld_dk) + h * DH + c4 * 4;
            *reinterpret_cast<uint2*>(p16) = f32x4_to_bf16(acc.x, acc.y, acc.z, acc.w);
            continue;
        }
        float* p = t ? a.dv + krow * a.ld_dv + h * DH + c4 * 4 : a.dk + krow * a.ld_dk + h * DH + c4 * 4;
        *reinterpret_cast<float4*>(p) = acc;
    }
}


int g_attn_bwd_rag_wave = 96;  // sola_tune "attn_bwd_rag_wave": ragged batches whose LONGEST unit has at most this many queries and keys take
                               // the per-wave kernels in their RAG form (0 = never)
int g_attn_bwd_blk = 1;    // sola_tune "attn_bwd_blk": 0 = per-wave staging (the round-1 kernels) for every shape (A/B)
int g_attn_bwd_small = 1;  // sola_tune "attn_bwd_small": 0 = two-pass kernels for every shape (A/B)
int g_attn_bwd_ablate = 0;  // measurement only (sola_tune "attn_bwd_ablate"): 1 = no tile arithmetic, 2 = only the first tile staged
int g_attn_bwd_fused = 1;  // sola_tune "attn_bwd_fused": 0 = two-pass kernels for the units of <= 128 queries and keys too (A/B)

int g_attn_bwd_bf16_mfma = 1;  // sola_tune "attn_bwd_bf16_mfma": 0 = the bf16-row launches keep the f32 products (bit-identical to the f32 kernel on the widened values)
template <int NWU, bool IO16, bool MF = false, bool G16 = false, bool O16 = false>
static int launch_bwd_fused_n(const AttnBwdArgs& a, int chunks, hipStream_t s) {
    constexpr int LD = 128 + 4;
    constexpr bool DMA = NWU == 4;  // as in the kernel
    constexpr size_t lds = ((size_t)(DMA ? 5 : 2) * 16 * LD + (NWU > 1 ? NWU * 16 * LD : 0) + NWU * 16 * 17 + (DMA ? 64 : 32)) * sizeof(float);
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<NWU, IO16, MF, G16, O16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.done(dev);
    }
    const long long blocks = (long long)a.G * a.H;
    SOLA_ARG(blocks < (1ll << 31) && chunks < 65536, "attention backward: grid too large");
    hipLaunchKernelGGL((attn_bwd_fused_kernel<NWU, IO16, MF, G16, O16>), dim3((unsigned)blocks, (unsigned)chunks), dim3(64 * NWU), lds, s, a);
    SOLA_LAUNCH_CHECK();
    if (a.qc_tiles) {
        // few (unit, head) pairs: the positions of a unit (at most 16 NWU keys x 64 float4) over several blocks
        const int ysplit = blocks >= 2048 ? 1 : std::min((16 * NWU * 64 + 255) / 256, (int)std::max<long long>(1, 2048 / blocks));
        hipLaunchKernelGGL(attn_bwd_part_reduce_kernel, dim3((unsigned)blocks, (unsigned)ysplit), dim3(256), 0, s, a, 16 * NWU);
        SOLA_LAUNCH_CHECK();
    }
    return SOLA_OK;
}
// queries per chunk of a chunked launch (object -> language: up to N * T' = 2048 queries against <= 64 keys): 16 tiles = 256
constexpr int kBwdChunkTiles = 16;
static bool bwd_fused_supported(const AttnBwdArgs& a, int DH, bool can_chunk) {
    if (!g_attn_bwd_fused || DH != 128 || a.Sk > 128) return false;
    return a.Sq <= 128 || (a.Sk <= 64 && (a.Sq <= 16 * kBwdChunkTiles || can_chunk));
}
// few-sample launches (at most kBwdFewRows query rows in total - one sample per step is 256-2000): chunks of 4 tiles = 64 queries, so that
// a single unit's eight (unit, head) blocks become 8 x Sq/64 (one sample of 256 queries: 75 -> ~25 us per launch)
constexpr int kBwdFewRows = 4096, kBwdFewChunkTiles = 4;
static int launch_bwd_fused(AttnBwdArgs a, bool can_chunk, long long part_rows, hipStream_t s) {
    // waves per (unit, head) = key tiles held in registers at a time, sized by the LONGEST unit of the launch
    int chunks = 1;
    a.qc_tiles = 0;
    a.ntile = g_attn_bwd_ablate;
    const int qc = (can_chunk && part_rows <= kBwdFewRows) ? kBwdFewChunkTiles : kBwdChunkTiles;
    if (a.Sq > 16 * qc && can_chunk) {
        a.qc_tiles = qc;
        chunks = ((a.Sq + 15) / 16 + qc - 1) / qc;
    }
    if (a.io16 && a.g16 && a.o16) {
        if (a.Sk <= 16) return launch_bwd_fused_n<1, true, true, true, true>(a, chunks, s);
        if (a.Sk <= 32) return launch_bwd_fused_n<2, true, true, true, true>(a, chunks, s);
        return launch_bwd_fused_n<4, true, true, true, true>(a, chunks, s);
    }
    if (a.io16 && a.g16) {  // (launch_attention_bwd: only with the bf16 products on)
        if (a.Sk <= 16) return launch_bwd_fused_n<1, true, true, true>(a, chunks, s);
        if (a.Sk <= 32) return launch_bwd_fused_n<2, true, true, true>(a, chunks, s);
        return launch_bwd_fused_n<4, true, true, true>(a, chunks, s);
    }
    if (a.io16 && g_attn_bwd_bf16_mfma) {
        if (a.Sk <= 16) return launch_bwd_fused_n<1, true, true>(a, chunks, s);
        if (a.Sk <= 32) return launch_bwd_fused_n<2, true, true>(a, chunks, s);
        return launch_bwd_fused_n<4, true, true>(a, chunks, s);
    }
    if (a.io16) {
        if (a.Sk <= 16) return launch_bwd_fused_n<1, true>(a, chunks, s);
        if (a.Sk <= 32) return launch_bwd_fused_n<2, true>(a, chunks, s);
        return launch_bwd_fused_n<4, true>(a, chunks, s);
    }
    if (a.Sk <= 16) return launch_bwd_fused_n<1, false>(a, chunks, s);
    if (a.Sk <= 32) return launch_bwd_fused_n<2, false>(a, chunks, s);
    return launch_bwd_fused_n<4, false>(a, chunks, s);
}

static int launch_bwd_small(const AttnBwdArgs& a, hipStream_t s) {
    const long long units = (long long)a.G * a.H;
    const unsigned blocks = (unsigned)((units + 7) / 8);
    const int need = a.Sq > a.Sk ? a.Sq : a.Sk;
    if (a.io16 && a.g16 && a.o16) {
        if (need <= 1) hipLaunchKernelGGL((attn_bwd_small_kernel<1, true, true, true>), dim3(blocks), dim3(256), 0, s, a);
        else if (need <= 2) hipLaunchKernelGGL((attn_bwd_small_kernel<2, true, true, true>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_bwd_small_kernel<4, true, true, true>), dim3(blocks), dim3(256), 0, s, a);
    } else if (a.io16 && a.g16) {
        if (need <= 1) hipLaunchKernelGGL((attn_bwd_small_kernel<1, true, true>), dim3(blocks), dim3(256), 0, s, a);
        else if (need <= 2) hipLaunchKernelGGL((attn_bwd_small_kernel<2, true, true>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_bwd_small_kernel<4, true, true>), dim3(blocks), dim3(256), 0, s, a);
    } else if (a.io16) {
        if (need <= 1) hipLaunchKernelGGL((attn_bwd_small_kernel<1, true>), dim3(blocks), dim3(256), 0, s, a);
        else if (need <= 2) hipLaunchKernelGGL((attn_bwd_small_kernel<2, true>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_bwd_small_kernel<4, true>), dim3(blocks), dim3(256), 0, s, a);
    } else if (need <= 1) hipLaunchKernelGGL((attn_bwd_small_kernel<1>), dim3(blocks), dim3(256), 0, s, a);
    else if (need <= 2) hipLaunchKernelGGL((attn_bwd_small_kernel<2>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_bwd_small_kernel<4>), dim3(blocks), dim3(256), 0, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

template <int DH>
int launch_bwd_dh(const AttnBwdArgs& a0, hipStream_t s) {
    constexpr size_t lds = (size_t)4 * 2 * 16 * (DH + 4) * sizeof(float);
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<DH>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<DH>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.done(dev);
    }
    AttnBwdArgs a = a0;
    bool done_dq = false;
    if (a.q_units && a.Sq <= g_attn_bwd_rag_wave && a.Sk <= g_attn_bwd_rag_wave) {
        // ragged batch of short units (motion attention over T' <= 25 steps, inter-object attention over N <= 80 tracks): one wave
        // per 16-row tile, no block-level sync, every wave runs its own unit's trip count.  A 64-row block of the shared-staging
        // shape would serve one unit of 3..80 rows with one to five of its tiles idle and all of them at the longest unit's pace.
        static DeviceOnce once_r;
        int dev_r;
        if (once_r.needed(&dev_r)) {
            SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<DH, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<DH, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            once_r.done(dev_r);
        }
        a.ntile = (a.Sq + 15) / 16;
        long long u = (long long)a.G * a.H * a.ntile;
        SOLA_ARG((u + 3) / 4 < (1ll << 31), "attention backward: grid too large");
        hipLaunchKernelGGL((attn_bwd_dq_kernel<DH, true>), dim3((unsigned)((u + 3) / 4)), dim3(256), lds, s, a);
        SOLA_LAUNCH_CHECK();
        a.ntile = (a.Sk + 15) / 16;
        u = (long long)a.G * a.H * a.ntile;
        SOLA_ARG((u + 3) / 4 < (1ll << 31), "attention backward: grid too large");
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, true>), dim3((unsigned)((u + 3) / 4)), dim3(256), lds, s, a);
        SOLA_LAUNCH_CHECK();
        return SOLA_OK;
    }
    if constexpr (DH == 128) if (g_attn_bwd_blk) {  // block-shared double-buffered staging (dvec is written by the dQ pass, read by the dK/dV pass)
        constexpr size_t lds2 = (size_t)2 * 2 * 16 * (DH + 4) * sizeof(float);
        const long long bq = (long long)a.G * a.H * ((a.Sq + 63) / 64), bk = (long long)a.G * a.H * ((a.Sk + 63) / 64);
        SOLA_ARG(bq < (1ll << 31) && bk < (1ll << 31), "attention backward: grid too large");
        // a block of the shared-staging shape serves 64 queries (dQ) / 64 keys (dK, dV) of one unit: with one 16-row tile per
        // unit three of its four waves would idle (16-step motion attention: 278 us per-wave vs 369 us), so those keep the
        // per-wave kernels, where a block is four units
        if (a.Sq > 16) {
            hipLaunchKernelGGL((attn_bwd_dq_blk_kernel<DH>), dim3((unsigned)bq), dim3(256), lds2, s, a);
            SOLA_LAUNCH_CHECK();
            done_dq = true;
        }
        if (a.Sk > 16) {
            if (!done_dq) {  // the dK/dV pass reads dvec = dO . O, which the dQ pass writes
                a.ntile = (a.Sq + 15) / 16;
                const long long u = (long long)a.G * a.H * a.ntile;
                hipLaunchKernelGGL((attn_bwd_dq_kernel<DH>), dim3((unsigned)((u + 3) / 4)), dim3(256), lds, s, a);
                SOLA_LAUNCH_CHECK();
                done_dq = true;
            }
            hipLaunchKernelGGL((attn_bwd_dkv_blk_kernel<DH>), dim3((unsigned)bk), dim3(256), lds2, s, a);
            SOLA_LAUNCH_CHECK();
            return SOLA_OK;
        }
    }
    long long units;
    if (!done_dq) {
        a.ntile = (a.Sq + 15) / 16;
        units = (long long)a.G * a.H * a.ntile;
        SOLA_ARG((units + 3) / 4 < (1ll << 31), "attention backward: grid too large");
        hipLaunchKernelGGL((attn_bwd_dq_kernel<DH>), dim3((unsigned)((units + 3) / 4)), dim3(256), lds, s, a);
        SOLA_LAUNCH_CHECK();
    }
    a.ntile = (a.Sk + 15) / 16;
    units = (long long)a.G * a.H * a.ntile;
    SOLA_ARG((units + 3) / 4 < (1ll << 31), "attention backward: grid too large");
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH>), dim3((unsigned)((units + 3) / 4)), dim3(256), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

}  // namespace

size_t attention_bwd_part_floats(long long q_rows, int G, int H, int Sk) {
    if (Sk > 64) return 0;
    const int qc = q_rows <= 4096 ? 4 : 16;  // kBwdFewRows / kBwdFewChunkTiles / kBwdChunkTiles (launch_bwd_fused)
    return (size_t)(q_rows / (16 * qc) + G + 1) * H * 2 * 64 * 128;
}

void sola_attn_set_bwd_small(int v) { g_attn_bwd_small = v; }
void sola_attn_set_bwd_blk(int v) { g_attn_bwd_blk = v; }
void sola_attn_set_bwd_rag_wave(int v) { g_attn_bwd_rag_wave = v; }
void sola_attn_set_bwd_fused(int v) { g_attn_bwd_fused = v; }
void sola_attn_set_bwd_ablate(int v) { g_attn_bwd_ablate = v; }
void sola_attn_set_bwd_bf16_mfma(int v) { g_attn_bwd_bf16_mfma = v; }
bool attention_bwd_dout_bf16_enabled() { return g_attn_bwd_bf16_mfma != 0; }

// bf16 q / k / v in and bf16 dQ / dK / dV out (AttnBwdDesc::io_bf16): the one-pass kernel's shapes, 8-value-aligned rows
bool attention_bwd_bf16_supported(const AttnBwdDesc& d) {
    if (d.DH != 128 || d.ldq % 8 || d.ldk % 8 || d.ldv % 8 || d.ld_dq % 4 || d.ld_dk % 4 || d.ld_dv % 4) return false;
    if (g_attn_bwd_small && d.Sq <= 4 && d.Sk <= 4) return true;  // the register kernel of sequences of <= 4 steps
    if (!g_attn_bwd_fused || d.Sk > 128) return false;
    const bool can_chunk = d.part && d.Sk <= 64 && d.part_floats >= attention_bwd_part_floats(d.part_rows, d.G, d.H, d.Sk) && (d.q_units || (d.q_rs == 1 && d.inner == 1));
    return d.Sq <= 128 || (d.Sk <= 64 && (d.Sq <= 16 * 16 || can_chunk));
}

int launch_attention_bwd(const AttnBwdDesc& d, hipStream_t s) {
    SOLA_ARG(d.G > 0 && d.H > 0 && d.Sq > 0 && d.Sk > 0 && d.inner > 0, "attention backward: bad sizes");
    AttnBwdArgs a;
    a.q = d.q; a.k = d.k; a.v = d.v; a.o = d.o; a.dout = d.dout; a.lse = d.lse;
    a.dq = d.dq; a.dk = d.dk; a.dv = d.dv; a.dvec = d.dvec;
    a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo;
    a.ld_dq = d.ld_dq; a.ld_dk = d.ld_dk; a.ld_dv = d.ld_dv;
    a.G = d.G; a.H = d.H; a.Sq = d.Sq; a.Sk = d.Sk; a.inner = d.inner; a.ntile = 1;
    a.q_outer = d.q_outer; a.q_inner = d.q_inner; a.q_rs = d.q_rs;
    a.k_outer = d.k_outer; a.k_inner = d.k_inner; a.k_rs = d.k_rs;
    a.scale = d.scale;
    a.drop = d.drop;
    a.q_units = d.q_units; a.k_units = d.q_units ? (d.k_units ? d.k_units : d.q_units) : nullptr;
    a.io16 = d.io_bf16 ? 1 : 0;
    a.g16 = d.dout_bf16 ? 1 : 0;
    a.o16 = d.o_bf16 ? 1 : 0;
    SOLA_ARG(!d.o_bf16 || d.dout_bf16, "attention backward: a bf16 O goes with a bf16 dO");
    SOLA_ARG(!d.dout_bf16 || (d.io_bf16 && g_attn_bwd_bf16_mfma && d.ldo % 8 == 0), "attention backward: a bf16 dO goes with bf16 q / k / v and the bf16 products");
    a.dq16 = static_cast<unsigned short*>(d.dq16);
    if (d.io_bf16) { a.dk = static_cast<float*>(d.dk16); a.dv = static_cast<float*>(d.dv16); }
    if (d.io_bf16) SOLA_ARG(attention_bwd_bf16_supported(d) && d.dq16 && d.dk16 && d.dv16 && d.dq, "attention backward: bf16 q / k / v and gradients need the one-pass kernel's shapes");
    const double elems = (double)d.G * d.H * d.DH;
    SolaProfScope prof(SOLA_PROF_ATTN_BWD, s, 14.0 * elems * d.Sq * d.Sk, 4.0 * elems * (5.0 * d.Sq + 4.0 * d.Sk));
    if (g_attn_bwd_small && d.DH == 128 && d.Sq <= 4 && d.Sk <= 4) return launch_bwd_small(a, s);
    // chunked launches address their partial sums by the unit's first row: units of row stride 1 in unit order (the caller's promise
    // when it hands over `part`), sized by attention_bwd_part_floats()
    a.qc_tiles = 0;
    a.part = d.part;
    const bool can_chunk = d.part && d.Sk <= 64 && d.part_floats >= attention_bwd_part_floats(d.part_rows, d.G, d.H, d.Sk) &&
                           (d.q_units || (d.q_rs == 1 && d.inner == 1));
    if (bwd_fused_supported(a, d.DH, can_chunk)) return launch_bwd_fused(a, can_chunk, d.part_rows, s);
    switch (d.DH) {
        case 128: return launch_bwd_dh<128>(a, s);
        case 64: return launch_bwd_dh<64>(a, s);
        case 32: return launch_bwd_dh<32>(a, s);
        case 16: return launch_bwd_dh<16>(a, s);
        default: sola_set_error("attention backward: head_dim %d unsupported", d.DH); return SOLA_ERR_ARG;
    }
}
