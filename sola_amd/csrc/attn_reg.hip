// Register-only shape of the exact-f32 attention core: no LDS, no barrier, every wave on its own.
//
// The LDS shapes (attn.hip, attn_simple.hip) stage a unit's K/V tile per block: load -> ds_write -> barrier -> MFMAs ->
// barrier, and what hides that chain is the number of blocks a CU can hold next to each other.  Here a wave owns one
// 16-query tile of one (group, head) unit and takes every MFMA operand straight from global memory in the register
// layout the instruction wants:
//   S^T = K Q^T : A = K[key = 16t + (lane & 15)][16i + 4g + c], B = Q[query = lane & 15][16i + 4g + c]  (g = lane >> 4):
//                 both are float4 loads of 64 contiguous bytes per row and instruction, one load feeds 4 MFMAs;
//   O^T = V^T P : the reduction slot g of step (t, r) is key 16t + 4g + r - exactly where S^T left the probabilities - and
//                 the 16 output rows of an MFMA may be ANY 16 head dims: lane x takes V[key][64h + 4 pi(x) + c] with
//                 pi(x) = 4 (x & 3) + (x >> 2), a float4 load in which the 16 lanes of a key read 256 contiguous bytes,
//                 and lane (query, g) ends with o[query][16 c' + 4g + {0..3}] for c' = 0..7, the layout the stores want
//                 (float4 or split-f16 pieces, 64 contiguous bytes per query and instruction).
// The waves of a unit (Sq / 16 of them, neighbours in one block) re-read its K and V, so those come from L1 / L2 after the
// first touch; the block order keeps a unit's waves and the heads of a group on one XCD.  HBM traffic stays the algorithmic
// q + k + v + o.  Keys are taken 64 at a time with the online softmax, so any Sk works; units of up to 64 keys take one pass.
// Exact f32 (v_mfma_f32_16x16x4_f32), strided groups or unit tables, optional log-sum-exp, f32 or split-f16 output.
// head_dim 128.  tools/attention.py:66-72.
#include "kernels.h"

namespace {

typedef _Float16 half4v __attribute__((ext_vector_type(4)));

struct AttnRArgs {
    const float *q, *k, *v;
    float* o;
    float* lse;
    int ldq, ldk, ldv, ldo;
    int G, H, Sq, Sk, inner, nqt;
    long long q_outer, q_inner, q_rs;
    long long k_outer, k_inner, k_rs;
    long long n_tasks;
    float scale;
    int o_sp16;
    int* guard;
    const int4 *q_units, *k_units;
};

__device__ __forceinline__ float xor16_32_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor16_32_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// One pass over up to 64 keys (NT tiles of 16) of a unit.  Two float4[8] buffers carry the K tiles and then the V tiles: the
// loads of tile t + 1 are issued before the MFMAs of tile t (sched_barrier keeps the compiler from sinking them to their
// uses - left alone it waits for every pair of loads in front of the eight MFMAs that need them), and the first V tile
// travels during the softmax.
struct RegState {
    f32x4 oacc[2][4];
    float m_run, l_run;
};
template <int NT>
__device__ __forceinline__ void reg_chunk(const float4 (&qf)[8], RegState& st, const char* kbase, const char* vbase, unsigned kstep,
                                          unsigned vstep, unsigned klane, unsigned vlane, int kb, int Sk, int x, int g4, float scale) {
    // kbase / vbase: wave-uniform byte pointers to the unit's first key row (head slice); a lane's address is a 32-bit byte
    // offset from it (scalar base + vector offset addressing: one VGPR per row instead of a 64-bit pointer)
    float4 buf[2][8];
    auto load_k = [&](float4 (&kf)[8], int t) {
        const int key = kb + 16 * t + x;
        const char* kp = kbase + ((unsigned)(key < Sk ? key : Sk - 1) * kstep + klane);
#pragma unroll
        for (int i = 0; i < 8; ++i) kf[i] = *reinterpret_cast<const float4*>(kp + 64 * i);
    };
    auto load_v = [&](float4 (&vf)[8], int t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kb + 16 * t + 4 * g4 + r;  // probabilities of keys past Sk are 0: any valid row will do
            const char* vp = vbase + ((unsigned)(key < Sk ? key : Sk - 1) * vstep + vlane);
            vf[2 * r] = *reinterpret_cast<const float4*>(vp);
            vf[2 * r + 1] = *reinterpret_cast<const float4*>(vp + 256);
        }
    };
    f32x4 sc[NT];
    load_k(buf[0], 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t + 1 < NT) load_k(buf[(t + 1) & 1], t + 1);
        else load_v(buf[NT & 1], 0);
        __builtin_amdgcn_sched_barrier(0);
        const float4 (&kf)[8] = buf[t & 1];
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i].x, qf[i].x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i + 1].x, qf[i + 1].x, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i].y, qf[i].y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i + 1].y, qf[i + 1].y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i].z, qf[i].z, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i + 1].z, qf[i + 1].z, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i].w, qf[i].w, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i + 1].w, qf[i + 1].w, a1, 0, 0, 0);
        }
        const int key0 = kb + 16 * t + 4 * g4;
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[t][r] = (key0 + r < Sk) ? (a0[r] + a1[r]) * scale : -INFINITY;
        __builtin_amdgcn_sched_barrier(0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][r]);
    mx = xor16_32_max(mx);
    const float m_new = fmaxf(st.m_run, mx);
    const float alpha = __expf(st.m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[t][r] = __expf(sc[t][r] - m_new);
            rs += sc[t][r];
        }
    rs = xor16_32_sum(rs);
    st.l_run = st.l_run * alpha + rs;
    st.m_run = m_new;
    if (kb > 0) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int c = 0; c < 4; ++c) st.oacc[hf][c] *= alpha;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t + 1 < NT) load_v(buf[(NT + t + 1) & 1], t + 1);
        __builtin_amdgcn_sched_barrier(0);
        const float4 (&vf)[8] = buf[(NT + t) & 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 v0 = vf[2 * r], v1 = vf[2 * r + 1];
            const float p = sc[t][r];
            st.oacc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.x, p, st.oacc[0][0], 0, 0, 0);
            st.oacc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.x, p, st.oacc[1][0], 0, 0, 0);
            st.oacc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.y, p, st.oacc[0][1], 0, 0, 0);
            st.oacc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.y, p, st.oacc[1][1], 0, 0, 0);
            st.oacc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.z, p, st.oacc[0][2], 0, 0, 0);
            st.oacc[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.z, p, st.oacc[1][2], 0, 0, 0);
            st.oacc[0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0.w, p, st.oacc[0][3], 0, 0, 0);
            st.oacc[1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1.w, p, st.oacc[1][3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int MINW>
__global__ __launch_bounds__(256, MINW) void attn_fwd_f32_reg_kernel(const AttnRArgs a) {
    constexpr int DH = 128;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: scalar address math
    const int x = lane & 15, g4 = lane >> 4;
    // consecutive logical blocks on one XCD (hardware deals blockIdx round-robin over the 8 XCDs); gridDim.x % 8 == 0
    const unsigned lb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long long task = (long long)lb * 4 + wave;
    if (task >= a.n_tasks) return;
    const long long unit = task / a.nqt;
    const int qt = (int)(task - unit * a.nqt);
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
    if (qt * 16 >= Sq || Sk <= 0) return;
    const int qi = qt * 16 + x;
    const bool q_ok = qi < Sq;
    const int qrow = q_ok ? qi : Sq - 1;

    float4 qf[8];
    {
        const float* qp = a.q + (q0 + (long long)qrow * q_rs) * a.ldq + h * DH + 4 * g4;
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[i] = *reinterpret_cast<const float4*>(qp + 16 * i);
    }
    // a unit's key rows span less than 4 GB (Sk * k_rs * ld * 4 bytes; checked at launch)
    const char* kbase = reinterpret_cast<const char*>(a.k + k0 * a.ldk + h * DH);
    const char* vbase = reinterpret_cast<const char*>(a.v + k0 * a.ldv + h * DH);
    const unsigned kstep = (unsigned)(k_rs * a.ldk) * 4u, vstep = (unsigned)(k_rs * a.ldv) * 4u;
    const unsigned klane = 16u * g4, vlane = 16u * (4 * (x & 3) + (x >> 2));

    RegState st;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int c = 0; c < 4; ++c) st.oacc[hf][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    st.m_run = -INFINITY; st.l_run = 0.f;

    for (int kb = 0; kb < Sk; kb += 64) {
        const int nt = (Sk - kb + 15) >> 4;  // wave-uniform
        if (nt >= 4) reg_chunk<4>(qf, st, kbase, vbase, kstep, vstep, klane, vlane, kb, Sk, x, g4, a.scale);
        else if (nt == 3) reg_chunk<3>(qf, st, kbase, vbase, kstep, vstep, klane, vlane, kb, Sk, x, g4, a.scale);
        else if (nt == 2) reg_chunk<2>(qf, st, kbase, vbase, kstep, vstep, klane, vlane, kb, Sk, x, g4, a.scale);
        else reg_chunk<1>(qf, st, kbase, vbase, kstep, vstep, klane, vlane, kb, Sk, x, g4, a.scale);
    }
    const float m_run = st.m_run, l_run = st.l_run;
    f32x4 (&oacc)[2][4] = st.oacc;
    if (!q_ok) return;
    const long long orow = q0 + (long long)qi * q_rs;
    if (a.lse && g4 == 0) a.lse[orow * a.H + h] = m_run + logf(l_run);
    const float inv = 1.f / l_run;
    float* op = a.o + orow * a.ldo + h * DH;
    // output row 4g + r' of tile (hf, c) is head dim 64 hf + 4 pi(4g + r') + c = 16 (4 hf + r') + 4g + c
    if (!a.o_sp16) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<float4*>(op + 16 * (4 * hf + r) + 4 * g4) =
                    make_float4(oacc[hf][0][r] * inv, oacc[hf][1][r] * inv, oacc[hf][2][r] * inv, oacc[hf][3][r] * inv);
        return;
    }
    // split-f16: the 8-wide block [hi8 | lo8] is shared by the lane pair (g4, g4 ^ 1); each lane writes the hi and the lo
    // halves of its own four values (block offset 8 * (g4 & 1), lo 16 bytes behind), as store_o in attn.hip
    float m = 0.f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            half4v hi, lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float v = oacc[hf][c][r] * inv;
                _Float16 h1, l1;
                split_f16(v, h1, l1);
                hi[c] = h1; lo[c] = l1;
                m = fmaxf(m, fabsf(v));
            }
            char* dst = reinterpret_cast<char*>(op + 16 * (4 * hf + r) + 8 * (g4 >> 1)) + 8 * (g4 & 1);
            *reinterpret_cast<half4v*>(dst) = hi;
            *reinterpret_cast<half4v*>(dst + 16) = lo;
        }
    if (a.guard && !(m < 65000.f)) atomicOr(a.guard, 1);
}

}  // namespace

int g_attn_reg = 1;       // sola_tune "attn_reg": 0 = never this shape, 2 = wherever it can run (A/B, tests)
int g_attn_reg_minw = 2;  // waves per SIMD the kernel is compiled for: 2 = 256 VGPRs, no spills (3 and 4 spill: 1.3x / 2.4x slower)
void sola_attn_set_reg(int v) { g_attn_reg = v; }
void sola_attn_set_reg_minw(int v) { g_attn_reg_minw = v; }

// f32 q / k / v at head_dim 128, no dropout; more than 4 queries or keys (attn_fwd_small_kernel has those).
// Where it wins (tools/attn_probe3.py, standalone launches, us):  motion attention of 16 steps (T = 128) 237 vs 292 for the
// packed LDS shape;  inter-object attention of 80 tracks 317 vs 389 (the LDS shapes restage a second, mostly empty K/V tile).
// Where it loses: 64 tracks 311 vs 279, 128 tracks 488 vs 414, object -> language (256 x 48) 250 vs 212 - every 16-query tile
// re-reads the unit's K and V from L2 with one tile of prefetch, which two waves per SIMD do not cover.
bool attention_reg_supported(const AttnDesc& d) {
    if (!g_attn_reg || d.drop.enabled || d.in_sp16 || d.DH != 128 || (d.Sq <= 4 && d.Sk <= 4)) return false;
    // the kernel addresses a unit's key rows with 32-bit byte offsets from the unit's first row
    const long long ld = d.ldk > d.ldv ? d.ldk : d.ldv;
    if (!d.q_units && (long long)d.Sk * d.k_rs * ld * 4 >= (1ll << 32)) return false;
    if (g_attn_reg == 2) return true;
    // ragged batches (unit tables) mix unit sizes in one launch: measured slower there (attention of the 128-sample MeViS-like
    // bench batch 2.73 -> 3.02 ms with this shape and attn_res.hip routed in), so they keep the high-occupancy LDS shape
    if (d.q_units) return false;
    return (d.Sq <= 16 && d.Sk <= 16) || (d.Sk > 64 && d.Sk <= 96);
}

int launch_attention_reg(const AttnDesc& d, hipStream_t s) {
    AttnRArgs a;
    a.q = d.q; a.k = d.k; a.v = d.v; a.o = d.o; a.lse = d.lse;
    a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo;
    a.G = d.G; a.H = d.H; a.Sq = d.Sq; a.Sk = d.Sk; a.inner = d.inner;
    a.q_outer = d.q_outer; a.q_inner = d.q_inner; a.q_rs = d.q_rs;
    a.k_outer = d.k_outer; a.k_inner = d.k_inner; a.k_rs = d.k_rs;
    a.scale = d.scale; a.o_sp16 = d.o_sp16; a.guard = d.o_sp16 ? d.guard : nullptr;
    a.q_units = d.q_units; a.k_units = d.q_units ? (d.k_units ? d.k_units : d.q_units) : nullptr;
    a.nqt = (d.Sq + 15) / 16;
    a.n_tasks = (long long)d.G * d.H * a.nqt;
    const long long blocks = ((a.n_tasks + 3) / 4 + 7) / 8 * 8;
    SOLA_ARG(blocks < (1ll << 31), "attention: grid too large");
    const double elems = (double)d.G * d.H * d.DH;
    SolaProfScope prof(SOLA_PROF_ATTN, s, 4.0 * elems * d.Sq * d.Sk, 4.0 * elems * (2.0 * d.Sq + 2.0 * d.Sk));
#ifdef SOLA_EXPERIMENTS  // the register budgets that spill (A/B record: 1.3x / 2.4x slower)
    if (g_attn_reg_minw >= 4) hipLaunchKernelGGL((attn_fwd_f32_reg_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else if (g_attn_reg_minw == 3) hipLaunchKernelGGL((attn_fwd_f32_reg_kernel<3>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else
#endif
    hipLaunchKernelGGL((attn_fwd_f32_reg_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
