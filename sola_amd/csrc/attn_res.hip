// Few keys, many queries (object -> language: N*T' = 256 ... 2048 tokens against L + 32 <= 64 text tokens,
// module/module.py:46-48): the unit's K and V (48 x 512 B each) are staged ONCE into LDS by an 8-wave block and stay there;
// every wave then streams 16-query tiles - Q straight from global memory in the MFMA operand layout, K / V fragments from
// LDS, O stored from registers - with nothing between the tiles but the loads themselves: no barrier after the staging one,
// no cross-unit state, ~120 VGPRs.  Two blocks (16 waves) share a CU, so four waves per SIMD take turns on the matrix pipe
// (82 us of exact-f32 MFMA per launch at the headline batch against ~115 us of HBM time at the practical rate) and cover each
// other's Q-load latency.  attn.hip's resident-K/V loop does the same arithmetic with 256 VGPRs (it carries the next unit's
// K/V through registers) at two 4-wave blocks per CU: 212 us per launch, 38 % of the HBM peak.
// Same register layout as attn.hip (S^T = K Q^T, O^T = V^T P^T, v_mfma_f32_16x16x4_f32), strided groups or unit tables,
// f32 or split-f16 output, optional log-sum-exp.  head_dim 128, at most 64 keys.  tools/attention.py:66-72.
#include "kernels.h"

namespace {

typedef _Float16 half4v __attribute__((ext_vector_type(4)));

struct AttnQArgs {
    const float *q, *k, *v;
    float* o;
    float* lse;
    int ldq, ldk, ldv, ldo;
    int G, H, Sq, Sk, inner, nchunk, tiles_per_wave;
    long long q_outer, q_inner, q_rs;
    long long k_outer, k_inner, k_rs;
    float scale;
    int o_sp16;
    int* guard;
    const int4 *q_units, *k_units;
    int k_priv;          // keys >= k_priv are the shared rows k_shared + (j - k_priv) (AttnDesc::k_private; INT_MAX = none)
    long long k_shared;
};

constexpr int RES_LD = 128 + 4;     // LDS row pitch in floats: 528 B = 16 B past two bank rows, so the 16 keys of a ds_read_b128
                                    // (K fragment) and the 4 keys of a ds_read_b32 (V^T fragment, keys 4 apart) hit distinct banks

template <int NT>  // key tiles of 16: Sk <= 16 NT
__device__ __forceinline__ void res_tile(const AttnQArgs& a, const float* Ks, const float* Vs, const float4 (&qf)[8], float* op, float* lsep,
                                         int Sk, int x, int g4, bool q_ok) {
    f32x4 sc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float* kr = Ks + (16 * t + x) * RES_LD + 4 * g4;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const float4 k0 = *reinterpret_cast<const float4*>(kr + 16 * i);
            const float4 k1 = *reinterpret_cast<const float4*>(kr + 16 * i + 16);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.x, qf[i].x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.x, qf[i + 1].x, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.y, qf[i].y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.y, qf[i + 1].y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.z, qf[i].z, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.z, qf[i + 1].z, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.w, qf[i].w, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.w, qf[i + 1].w, a1, 0, 0, 0);
        }
        const int key0 = 16 * t + 4 * g4;
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[t][r] = (key0 + r < Sk) ? (a0[r] + a1[r]) * a.scale : -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
    mx = max_xor32(max_xor16(mx));
    float rs = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[t][r] = __expf(sc[t][r] - mx);
            rs += sc[t][r];
        }
    rs = sum_xor32(sum_xor16(rs));
    f32x4 oacc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vp = Vs + (16 * t + 4 * g4 + r) * RES_LD + x;
#pragma unroll
            for (int c = 0; c < 8; ++c) oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[16 * c], sc[t][r], oacc[c], 0, 0, 0);
        }
    if (!q_ok) return;
    if (lsep && g4 == 0) *lsep = mx + logf(rs);
    const float inv = 1.f / rs;
    if (!a.o_sp16) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            *reinterpret_cast<float4*>(op + 16 * c + 4 * g4) = make_float4(oacc[c][0] * inv, oacc[c][1] * inv, oacc[c][2] * inv, oacc[c][3] * inv);
        return;
    }
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        half4v hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = oacc[c][j] * inv;
            _Float16 h1, l1;
            split_f16(v, h1, l1);
            hi[j] = h1; lo[j] = l1;
            m = fmaxf(m, fabsf(v));
        }
        char* dst = reinterpret_cast<char*>(op + 16 * c + 8 * (g4 >> 1)) + 8 * (g4 & 1);
        *reinterpret_cast<half4v*>(dst) = hi;
        *reinterpret_cast<half4v*>(dst + 16) = lo;
    }
    if (a.guard && !(m < 65000.f)) atomicOr(a.guard, 1);
}

// ---- the same shape in the split precision mode's arithmetic (AttnDesc::split_math) -----------------------------------------
// The exact-f32 MFMA makes this shape co-bound by the matrix pipe (82 us of v_mfma_f32_16x16x4_f32 per launch at the headline
// batch next to ~115 us of HBM time; measured 163-182 us).  In the split precision mode every dense contraction already runs
// as hi*hi + hi*lo + lo*hi on f16 MFMAs (22-bit products), and here the conversion is cheap because K and V are staged ONCE
// per block for 256 ... 2048 queries: while staging, K becomes (hi, lo) f16 rows and V becomes (hi, lo) TRANSPOSED f16 rows
// ([dh][key], so both MFMA A-fragments are 8-byte LDS reads); per 16-query tile Q (32 values per lane) and P (12) are split
// in registers.  3 x v_mfma_f32_16x16x16_f16 per 16-deep product step: 144 MFMAs of 8 cycles per tile instead of 192 of 32.
// (attn_simple.hip's first attempt converted K/V per 64-query block and every wave re-split the fragments it read: slower.)
typedef _Float16 half4r __attribute__((ext_vector_type(4)));
struct HL4r { half4r hi, lo; };
__device__ __forceinline__ HL4r split4r(float x, float y, float z, float w, float& amax) {
    HL4r r;
    const float in[4] = {x, y, z, w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        _Float16 h1, l1;
        split_f16(in[j], h1, l1);
        r.hi[j] = h1; r.lo[j] = l1;
        amax = fmaxf(amax, fabsf(in[j]));
    }
    return r;
}
__device__ __forceinline__ f32x4 mfma3r(const HL4r& a, const HL4r& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a.hi, b.hi, c, 0, 0, 0);
}
constexpr int SM_KP = 136;  // K row pitch in halfs: 272 B = 16 B past a bank row, the 8-byte fragment reads of 16 keys x 2 k-groups tile the banks
constexpr int SM_VP = 72;   // V^T row pitch in halfs: 144 B = 9 x 16 B (odd), up to 64 keys per row

template <int NT>
__device__ __forceinline__ void sm_tile(const AttnQArgs& a, const _Float16* Kh, const _Float16* Kl, const _Float16* Vh, const _Float16* Vl,
                                        const float4 (&qf)[8], float* op, float* lsep, int Sk, int x, int g4, bool q_ok, float& amax) {
    HL4r q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = split4r(qf[i].x, qf[i].y, qf[i].z, qf[i].w, amax);
    f32x4 sc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ko = (16 * t + x) * SM_KP + 4 * g4;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            HL4r k0, k1;
            k0.hi = *reinterpret_cast<const half4r*>(Kh + ko + 16 * i); k0.lo = *reinterpret_cast<const half4r*>(Kl + ko + 16 * i);
            k1.hi = *reinterpret_cast<const half4r*>(Kh + ko + 16 * i + 16); k1.lo = *reinterpret_cast<const half4r*>(Kl + ko + 16 * i + 16);
            a0 = mfma3r(k0, q[i], a0);
            a1 = mfma3r(k1, q[i + 1], a1);
        }
        const int key0 = 16 * t + 4 * g4;
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[t][r] = (key0 + r < Sk) ? (a0[r] + a1[r]) * a.scale : -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
    mx = max_xor32(max_xor16(mx));
    float rs = 0.f;
    HL4r p[NT];
    float dummy = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[t][r] = __expf(sc[t][r] - mx);
            rs += sc[t][r];
        }
        p[t] = split4r(sc[t][0], sc[t][1], sc[t][2], sc[t][3], dummy);
    }
    rs = sum_xor32(sum_xor16(rs));
    f32x4 oacc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int vo = (16 * c + x) * SM_VP + 16 * t + 4 * g4;
            HL4r v;
            v.hi = *reinterpret_cast<const half4r*>(Vh + vo); v.lo = *reinterpret_cast<const half4r*>(Vl + vo);
            oacc[c] = mfma3r(v, p[t], oacc[c]);
        }
    if (!q_ok) return;
    if (lsep && g4 == 0) *lsep = mx + logf(rs);
    const float inv = 1.f / rs;
    if (!a.o_sp16) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            *reinterpret_cast<float4*>(op + 16 * c + 4 * g4) = make_float4(oacc[c][0] * inv, oacc[c][1] * inv, oacc[c][2] * inv, oacc[c][3] * inv);
        return;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const HL4r o = split4r(oacc[c][0] * inv, oacc[c][1] * inv, oacc[c][2] * inv, oacc[c][3] * inv, amax);
        char* dst = reinterpret_cast<char*>(op + 16 * c + 8 * (g4 >> 1)) + 8 * (g4 & 1);
        *reinterpret_cast<half4r*>(dst) = o.hi;
        *reinterpret_cast<half4r*>(dst + 16) = o.lo;
    }
}

#ifdef SOLA_EXPERIMENTS  // the object->language shape on f16-MFMA triples (sola_tune "attn_res_splitm"): measured no faster than exact f32, spills 20 registers
__global__ __launch_bounds__(512, 4) void attn_fwd_sm_res_kernel(const AttnQArgs a) {
    constexpr int DH = 128, NW = 8;
    extern __shared__ __attribute__((aligned(16))) float smem_r[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 15, g4 = lane >> 4;
    const unsigned lb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long long unit = lb / a.nchunk;
    if (unit >= (long long)a.G * a.H) return;
    const int chunk = (int)(lb - unit * a.nchunk);
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
    const int qchunk = NW * 16 * a.tiles_per_wave;
    const int qbeg = chunk * qchunk;
    if (qbeg >= Sq || Sk <= 0) return;  // block-uniform
    const int rows = (Sk + 15) & ~15;
    _Float16* Kh = reinterpret_cast<_Float16*>(smem_r);
    _Float16* Kl = Kh + rows * SM_KP;
    _Float16* Vh = Kl + rows * SM_KP;      // V^T: [dh][key]
    _Float16* Vl = Vh + DH * SM_VP;
    float amax = 0.f;
    // stage + convert: 32 float4 per row; rows past Sk are zero (a NaN there would survive the multiplication by p = 0)
    for (int e = tid; e < rows * 32; e += NW * 64) {
        const int r = e >> 5, c = e & 31;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (r < Sk) {
            const long long row = r < a.k_priv ? k0 + (long long)r * k_rs : a.k_shared + (r - a.k_priv);
            kv = *reinterpret_cast<const float4*>(a.k + row * a.ldk + h * DH + 4 * c);
            vv = *reinterpret_cast<const float4*>(a.v + row * a.ldv + h * DH + 4 * c);
        }
        const HL4r kk = split4r(kv.x, kv.y, kv.z, kv.w, amax), vs = split4r(vv.x, vv.y, vv.z, vv.w, amax);
        *reinterpret_cast<half4r*>(Kh + r * SM_KP + 4 * c) = kk.hi;
        *reinterpret_cast<half4r*>(Kl + r * SM_KP + 4 * c) = kk.lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Vh[(4 * c + j) * SM_VP + r] = vs.hi[j];
            Vl[(4 * c + j) * SM_VP + r] = vs.lo[j];
        }
    }
    __syncthreads();
    const int nt = rows >> 4;
    for (int j = 0; j < a.tiles_per_wave; ++j) {
        const int qt0 = qbeg + (j * NW + wave) * 16;  // the block's waves take neighbouring tiles
        if (qt0 >= Sq) break;
        const int qi = qt0 + x;
        const bool q_ok = qi < Sq;
        const long long qrow = q0 + (long long)(q_ok ? qi : Sq - 1) * q_rs;
        const float* qp = a.q + qrow * a.ldq + h * DH + 4 * g4;
        float4 qf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[i] = *reinterpret_cast<const float4*>(qp + 16 * i);
        float* op = a.o + qrow * a.ldo + h * DH;
        float* lsep = a.lse ? a.lse + qrow * a.H + h : nullptr;
        if (nt == 3) sm_tile<3>(a, Kh, Kl, Vh, Vl, qf, op, lsep, Sk, x, g4, q_ok, amax);
        else if (nt == 4) sm_tile<4>(a, Kh, Kl, Vh, Vl, qf, op, lsep, Sk, x, g4, q_ok, amax);
        else if (nt == 2) sm_tile<2>(a, Kh, Kl, Vh, Vl, qf, op, lsep, Sk, x, g4, q_ok, amax);
        else sm_tile<1>(a, Kh, Kl, Vh, Vl, qf, op, lsep, Sk, x, g4, q_ok, amax);
    }
    // anything that left the f16 range (q, k, v or the output pairs; NaN fails the comparison too): the forward repeats in f32
    if (a.guard && !(amax < 65000.f)) atomicOr(a.guard, 1);
}
#endif

// NW waves per block, compiled for MINW waves per SIMD.  PF: the Q rows of a wave's next tile are requested into a second
// register set before the current tile is computed (needs the 168 VGPRs of MINW = 3; at 128 it spills and loses).
template <int NW, int MINW, bool PF>
__global__ __launch_bounds__(NW * 64, MINW) void attn_fwd_f32_res_kernel(const AttnQArgs a) {
    constexpr int DH = 128;
    extern __shared__ __attribute__((aligned(16))) float smem_r[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 15, g4 = lane >> 4;
    // consecutive logical blocks (the chunks of a unit, the heads of a group) on one XCD; gridDim.x % 8 == 0
    const unsigned lb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long long unit = lb / a.nchunk;
    if (unit >= (long long)a.G * a.H) return;
    const int chunk = (int)(lb - unit * a.nchunk);
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
    const int qchunk = NW * 16 * a.tiles_per_wave;
    const int qbeg = chunk * qchunk;
    if (qbeg >= Sq || Sk <= 0) return;  // block-uniform
    // tile j of this wave: the block's waves take neighbouring tiles
    auto tile_live = [&](int j) { return j < a.tiles_per_wave && qbeg + (j * NW + wave) * 16 < Sq; };  // wave-uniform
    auto q_row = [&](int j) {
        const int qi = qbeg + (j * NW + wave) * 16 + x;
        return q0 + (long long)(qi < Sq ? qi : Sq - 1) * q_rs;
    };
    auto load_q = [&](float4 (&qf)[8], int j) {
        const float* qp = a.q + q_row(j) * a.ldq + h * DH + 4 * g4;
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[i] = *reinterpret_cast<const float4*>(qp + 16 * i);
    };
    float4 qA[8], qB[8];
    if (PF && tile_live(0)) load_q(qA, 0);  // travels while K/V are staged
    const int rows = (Sk + 15) & ~15;
    float* Ks = smem_r;
    float* Vs = smem_r + rows * RES_LD;
    // stage K and V: 32 float4 per row; rows past Sk are zero (a NaN there would survive the multiplication by p = 0)
    for (int e = tid; e < rows * 32; e += NW * 64) {
        const int r = e >> 5, c = e & 31;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (r < Sk) {
            const long long row = r < a.k_priv ? k0 + (long long)r * k_rs : a.k_shared + (r - a.k_priv);
            kv = *reinterpret_cast<const float4*>(a.k + row * a.ldk + h * DH + 4 * c);
            vv = *reinterpret_cast<const float4*>(a.v + row * a.ldv + h * DH + 4 * c);
        }
        *reinterpret_cast<float4*>(Ks + r * RES_LD + 4 * c) = kv;
        *reinterpret_cast<float4*>(Vs + r * RES_LD + 4 * c) = vv;
    }
    __syncthreads();
    const int nt = rows >> 4;
    auto compute = [&](const float4 (&qf)[8], int j) {
        const int qi = qbeg + (j * NW + wave) * 16 + x;
        const long long qrow = q_row(j);
        float* op = a.o + qrow * a.ldo + h * DH;
        float* lsep = a.lse ? a.lse + qrow * a.H + h : nullptr;
        const bool q_ok = qi < Sq;
        if (nt == 3) res_tile<3>(a, Ks, Vs, qf, op, lsep, Sk, x, g4, q_ok);
        else if (nt == 4) res_tile<4>(a, Ks, Vs, qf, op, lsep, Sk, x, g4, q_ok);
        else if (nt == 2) res_tile<2>(a, Ks, Vs, qf, op, lsep, Sk, x, g4, q_ok);
        else res_tile<1>(a, Ks, Vs, qf, op, lsep, Sk, x, g4, q_ok);
    };
    if (PF) {
        for (int j = 0; tile_live(j); j += 2) {
            if (tile_live(j + 1)) load_q(qB, j + 1);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch in front of the tile's MFMAs (the scheduler sinks loads to their uses)
            compute(qA, j);
            if (!tile_live(j + 1)) break;
            if (tile_live(j + 2)) load_q(qA, j + 2);
            __builtin_amdgcn_sched_barrier(0);
            compute(qB, j + 1);
        }
    } else {
        for (int j = 0; tile_live(j); ++j) {
            load_q(qA, j);
            compute(qA, j);
        }
    }
}

}  // namespace

int g_attn_res = 1;              // sola_tune "attn_res": 0 = never this shape (A/B), 2 = also for ragged batches
int g_attn_res_tiles = 0;        // 16-query tiles per wave and block; 0 = up to 4, the chunks of a unit made equal
int g_attn_res_splitm = 0;      // sola_tune "attn_res_splitm": 1 = f16-MFMA triples when the caller's arithmetic is the split mode's.  Measured:
                                // 181 -> 175 us at 256 x 48, 163 -> 165 us at 2048 x 48 - removing 3/4 of the matrix-pipe time buys almost
                                // nothing, so the pipe is not what bounds this shape (its 64-byte pieces per row and instruction are); off
int g_attn_res_shape = 0;        // 0 = auto, 1 = 8 waves / 128 VGPRs / no prefetch, 2 = 4 waves / 168 VGPRs / next-tile prefetch
void sola_attn_set_res(int v) { g_attn_res = v; }
void sola_attn_set_res_tiles(int v) { g_attn_res_tiles = v < 0 ? 0 : v; }
void sola_attn_set_res_shape(int v) { g_attn_res_shape = v; }
void sola_attn_set_res_splitm(int v) { g_attn_res_splitm = v; }

// f32 q / k / v at head_dim 128, no dropout, at most 64 keys and enough queries per unit to pay for staging K/V
bool attention_res_supported(const AttnDesc& d) {
    if (d.q_units && g_attn_res != 2) return false;  // ragged batches: see attention_reg_supported
    return g_attn_res && !d.drop.enabled && !d.in_sp16 && d.DH == 128 && d.Sk <= 64 && d.Sq >= 128;
}

bool attention_shared_keys_supported(const AttnDesc& d) { return attention_res_supported(d) && !d.q_units && d.k_private > 0 && d.k_private <= d.Sk; }

template <int NW, int MINW, bool PF>
static int launch_res(AttnQArgs a, const AttnDesc& d, hipStream_t s) {
    const int max_tiles = PF ? 8 : 4;
    if (g_attn_res_tiles > 0) {
        a.tiles_per_wave = g_attn_res_tiles;
    } else {
        const int per_tile = NW * 16, nch = (d.Sq + max_tiles * per_tile - 1) / (max_tiles * per_tile);
        a.tiles_per_wave = (d.Sq + per_tile * nch - 1) / (per_tile * nch);
    }
    const int qchunk = NW * 16 * a.tiles_per_wave;
    a.nchunk = (d.Sq + qchunk - 1) / qchunk;
    const long long blocks = ((long long)d.G * d.H * a.nchunk + 7) / 8 * 8;
    SOLA_ARG(blocks < (1ll << 31), "attention: grid too large");
    const size_t lds = (size_t)2 * ((d.Sk + 15) & ~15) * RES_LD * sizeof(float);
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_f32_res_kernel<NW, MINW, PF>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 64 * RES_LD * sizeof(float))));
        once.done(dev);
    }
    hipLaunchKernelGGL((attn_fwd_f32_res_kernel<NW, MINW, PF>), dim3((unsigned)blocks), dim3(NW * 64), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_attention_res(const AttnDesc& d, hipStream_t s) {
    AttnQArgs a;
    a.q = d.q; a.k = d.k; a.v = d.v; a.o = d.o; a.lse = d.lse;
    a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo;
    a.G = d.G; a.H = d.H; a.Sq = d.Sq; a.Sk = d.Sk; a.inner = d.inner;
    a.q_outer = d.q_outer; a.q_inner = d.q_inner; a.q_rs = d.q_rs;
    a.k_outer = d.k_outer; a.k_inner = d.k_inner; a.k_rs = d.k_rs;
    a.scale = d.scale; a.o_sp16 = d.o_sp16; a.guard = d.o_sp16 ? d.guard : nullptr;
    a.q_units = d.q_units; a.k_units = d.q_units ? (d.k_units ? d.k_units : d.q_units) : nullptr;
    a.tiles_per_wave = 1; a.nchunk = 1;
    a.k_priv = d.k_private > 0 ? d.k_private : 0x7fffffff;
    a.k_shared = d.k_shared_row;
    const double elems = (double)d.G * d.H * d.DH;
    SolaProfScope prof(SOLA_PROF_ATTN, s, 4.0 * elems * d.Sq * d.Sk, 4.0 * elems * (2.0 * d.Sq + 2.0 * d.Sk));
    // measured at 256 samples (256 queries x 48 keys per unit, tools/attn_probe3.py): 8 waves without prefetch 180-183 us, three
    // 4-wave blocks per CU with the next tile's Q prefetched (pinned in front of the MFMAs by sched_barrier) 191-193 us,
    // attn.hip's resident loop 205 us.  Also measured without effect: staggered block starts (0-8 us), f16-MFMA triples (175 us:
    // a quarter of the matrix-pipe time), and a plain copy with this kernel's 16-rows-x-64-bytes instruction footprint runs at
    // 5.2-5.8 TB/s (tools/micro/strided_bw) - so neither the pipe, nor the bytes in flight, nor the access pattern, nor
    // phase-locking explains the 3.6 TB/s.  A per-phase timeline (wall_clock64 stamps per wave, 256 samples) reads: 7.0 us to the
    // staging barrier, 4.7 us for the first tile's Q rows, 9.3 us of MFMA phase, 1.4 us of store drain, then 2.5 + 6.3 + 1.0 us
    // for the second tile - about half of a 32 us block is memory latency under load.  Acting on it did not pay: requesting
    // the first tile's Q rows before the staging 183 -> 197 us (the K/V loads queue behind them), all staging loads in front of
    // the first LDS write 183 -> 183 us
#ifdef SOLA_EXPERIMENTS
    if (d.split_math && g_attn_res_splitm && d.k_private <= 0) {  // the split precision mode's arithmetic (f16 MFMA triples, K/V converted once per block)
        a.guard = d.guard;
        a.tiles_per_wave = g_attn_res_tiles > 0 ? g_attn_res_tiles : 0;
        if (a.tiles_per_wave == 0) {
            const int per_tile = 8 * 16, nch = (d.Sq + 4 * per_tile - 1) / (4 * per_tile);
            a.tiles_per_wave = (d.Sq + per_tile * nch - 1) / (per_tile * nch);
        }
        const int qchunk = 8 * 16 * a.tiles_per_wave;
        a.nchunk = (d.Sq + qchunk - 1) / qchunk;
        const long long blocks = ((long long)d.G * d.H * a.nchunk + 7) / 8 * 8;
        SOLA_ARG(blocks < (1ll << 31), "attention: grid too large");
        const int rows = (d.Sk + 15) & ~15;
        const size_t lds = ((size_t)2 * rows * SM_KP + (size_t)2 * 128 * SM_VP) * sizeof(_Float16);
        static DeviceOnce once;
        int dev;
        if (once.needed(&dev)) {
            SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_sm_res_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(((size_t)2 * 64 * SM_KP + (size_t)2 * 128 * SM_VP) * sizeof(_Float16))));
            once.done(dev);
        }
        hipLaunchKernelGGL(attn_fwd_sm_res_kernel, dim3((unsigned)blocks), dim3(512), lds, s, a);
        SOLA_LAUNCH_CHECK();
        return SOLA_OK;
    }
#endif
    if (g_attn_res_shape == 2) return launch_res<4, 3, true>(a, d, s);
    return launch_res<8, 4, false>(a, d, s);
}
