// Ring-staged shape of the exact-f32 attention core (inference, head_dim 128; round 5).
// A CLOSED EXPERIMENT: compiled in SOLA_EXPERIMENTS builds only (make EXPERIMENTS=1; sola_tune "attn_ring" 1, default 0), kept as the
// instrument behind profiles/r05_attention_ring.txt.  Bit-identical to attn_simple.hip's kernel on every shape tried, and SLOWER in
// steady state (inter-object attention at the headline dims: 239.8 vs 229.6 us; N = 100, T' = 8: 640 vs 586; T = 128 stress: 370 vs
// 363).  Its ablations say why nothing at this site moves any more: with all arithmetic removed (DMA, barriers and stores only) the
// launch still takes 215 us = 5.0 TB/s at one OR two blocks per CU - the memory system's rate for 512-byte pieces at a 16 KB
// stride with a quarter of the bytes written - and with all loads removed 181 us; the shipped kernel's 229 us sits 6 % above that floor.
//
// attn_simple.hip's high-occupancy shape keeps ONE 16-key tile per block in flight (16 KB in registers on their way to LDS) and
// nothing across units: a block loads q, waits, walks its four tiles and leaves - 4.2 TB/s at the inter-object attention, with the
// matrix pipe 44 % busy and the memory system 53 %: neither saturated, the chip waits on latency.  Here a block is PERSISTENT and
// its K/V stream never stops:
//   * K and V tiles (16 keys x 512 B each) travel global -> LDS by DMA (buffer_load_dwordx4 ... lds: no registers in flight) into a
//     ring of three 16 KB stages; the producer cursor runs two tiles ahead of the consumer ACROSS work items (a work item = one
//     64-query block of one (group, head) unit), so the loads of the next unit are in flight while this one is multiplied;
//   * a wave's 16 query rows arrive the same way in a wave-private 8 KB tile, requested one work item ahead (right behind the
//     fragment reads of the current item);
//   * rows are 512 contiguous bytes with their 16-byte chunks XOR-swizzled by the row on the SOURCE address (the DMA writes a
//     wave's 64 x 16 B linearly): K / q fragments (ds_read_b128, row = lane % 16) and V elements (ds_read2st64_b32) are conflict-free
//     without padding;
//   * rows past a unit's end are sent out of the descriptor's range: the DMA writes zeros (tools/micro/lds_dma_oob.hip);
//   * the waits are manual: the fragment reads are inline asm (a compiler-visible LDS read behind an LDS-DMA gets vmcnt(0), which
//     would drain the ring at every tile - gemm_f32p.hip), s_waitcnt vmcnt(n) with n = the vector-memory operations this wave issued
//     behind the tile it needs (the counter decrements in order), one barrier per tile.
// 80 KB of LDS, two blocks per CU: 64-96 KB per CU in flight at any time (8 TB/s x ~2 us of loaded latency / 256 CUs = 62 KB).
// Arithmetic: attn_fwd_f32_simple_kernel<128, 16>'s, operand for operand (v_mfma_f32_16x16x4_f32, S^T = K Q^T / O^T = V^T P^T, online
// softmax per 16-key tile): the outputs are bit-identical to that kernel's.
#include <type_traits>

#include "../kernels.h"

#ifdef SOLA_EXPERIMENTS
namespace {

typedef __attribute__((address_space(3))) void* lptr_t;
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4s __attribute__((ext_vector_type(4)));

struct AttnRArgs {
    const float *q, *k, *v;
    float* o;
    int ldq, ldk, ldv, ldo;
    int G, H, Sq, Sk, inner, nqb;
    long long q_outer, q_inner, q_rs;
    long long k_outer, k_inner, k_rs;
    float scale;
    int o_sp16;
    int* guard;
    const int4 *q_units, *k_units;
    int n_items;
    int xcd_remap;
};

constexpr int RDH = 128;
constexpr int RROWB = RDH * 4;          // a head slice of a row: 512 bytes = 32 chunks of 16
constexpr int RKV = 16 * RROWB;         // the K (or V) half of a stage: 16 keys
constexpr int RSTAGE = 2 * RKV;         // 16 KB
constexpr int RNST = 3;
constexpr int RQ_OFF = RNST * RSTAGE;   // the waves' private query tiles sit behind the ring
constexpr int RQ_WAVE = 16 * RROWB;     // 8 KB
constexpr int RLDS = RQ_OFF + 4 * RQ_WAVE;  // 80 KB

#define RING_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// ABL (measurement, SOLA_EXPERIMENTS builds only; results are garbage): 1 = no fragment reads / products / softmax, 2 = no DMA
template <int ABL = 0>
__global__ __launch_bounds__(256, 2) void attn_fwd_f32_ring_kernel(const AttnRArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr unsigned OOB = 0x80000000u;  // beyond every descriptor's range: the DMA writes zeros
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;
    const int hi = lane >> 5, lo5 = lane & 31;

    struct Item {
        long long q0, k0;
        int q_rs, k_rs, Sq, Sk, qb, h, grp;
    };
    const int stride = (int)gridDim.x, n_items = a.n_items;
    // logical block order: with xcd_remap consecutive logical blocks - the q-blocks of a unit, the heads of a group - share an XCD's L2
    const int lb = a.xcd_remap ? (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    auto decode = [&](int idx, Item& it) -> bool {
        const int unit = idx / a.nqb;
        it.qb = idx - unit * a.nqb;
        it.grp = unit / a.H;
        it.h = unit - it.grp * a.H;
        if (a.q_units) {
            // scalar loads by hand: a compiler-visible vector load would be waited for with vmcnt(0) - the whole ring drained per item
            i32x4s qu, ku;
            asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(qu), "=&s"(ku)
                         : "s"(a.q_units + it.grp), "s"(a.k_units + it.grp)
                         : "memory");
            it.q0 = qu[0]; it.q_rs = qu[1]; it.Sq = qu[2];
            it.k0 = ku[0]; it.k_rs = ku[1]; it.Sk = ku[2];
        } else {
            it.q0 = (long long)(it.grp / a.inner) * a.q_outer + (long long)(it.grp % a.inner) * a.q_inner;
            it.k0 = (long long)(it.grp / a.inner) * a.k_outer + (long long)(it.grp % a.inner) * a.k_inner;
            it.q_rs = (int)a.q_rs; it.k_rs = (int)a.k_rs; it.Sq = a.Sq; it.Sk = a.Sk;
        }
        return it.qb * 64 < it.Sq;  // ragged: fewer q-blocks than the largest unit
    };
    auto next_valid = [&](int& idx, Item& it) -> bool {
        while (idx < n_items) {
            if (decode(idx, it)) return true;
            idx += stride;
        }
        return false;
    };

    int issued = 0;  // vector-memory operations this wave has issued (a lower bound: an uncounted one only strengthens a wait)
    int mark[RNST] = {0, 0, 0}, mark_q = 0;
    // wait until at most n of this wave's vector-memory operations are outstanding (n rounded down to a multiple of four)
    auto wait_vm = [&](int n) {
        switch (n >> 2) {
            case 0: RING_VM(0); break;
            case 1: RING_VM(4); break;
            case 2: RING_VM(8); break;
            case 3: RING_VM(12); break;
            case 4: RING_VM(16); break;
            case 5: RING_VM(20); break;
            case 6: RING_VM(24); break;
            case 7: RING_VM(28); break;
            case 8: RING_VM(32); break;
            case 9: RING_VM(36); break;
            default: RING_VM(40); break;
        }
    };

    // ---- query stream: this wave's 16 rows of a work item, 8 DMA instructions of two rows; lane -> (row 2 i + lane / 32, chunk)
    auto issue_q = [&](const Item& it) {
        const long long row0 = it.q0 + (long long)(it.qb * 64 + wave * 16) * it.q_rs;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.q) + row0 * a.ldq + it.h * RDH, 0, 0x7fffff00, 0x00020000);
        const int rowb = it.q_rs * a.ldq * 4;
        const int left = it.Sq - (it.qb * 64 + wave * 16);  // valid rows of this wave's tile (<= 0: none)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 2 * i + hi;
            unsigned vo = (unsigned)(hi * rowb + ((lo5 ^ r) << 4));
            if (left < 16) vo = r < left ? vo : OOB;
            if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + RQ_OFF + wave * RQ_WAVE + i * 1024), 16, vo, 2 * i * rowb, 0, 0);
        }
        if (!(ABL & 2)) issued += 8;
        mark_q = issued;
    };

    // ---- K / V stream: a stage = 16 keys of K then of V; this wave owns rows 4 w .. 4 w + 3 of both (two instructions each)
    Item p_it;
    int p_idx = lb;
    bool p_live = next_valid(p_idx, p_it);
    if (!p_live) return;
    int p_kt = 0;
    __amdgpu_buffer_rsrc_t rs_k, rs_v;
    int p_krowb = 0, p_vrowb = 0;
    unsigned kvo[2], vvo[2];
    auto setup_p = [&]() {
        const long long row0 = p_it.k0;
        rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.k) + row0 * a.ldk + p_it.h * RDH, 0, 0x7fffff00, 0x00020000);
        rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.v) + row0 * a.ldv + p_it.h * RDH, 0, 0x7fffff00, 0x00020000);
        p_krowb = p_it.k_rs * a.ldk * 4;
        p_vrowb = p_it.k_rs * a.ldv * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 4 * wave + 2 * i + hi;
            kvo[i] = (unsigned)(r * p_krowb + ((lo5 ^ r) << 4));
            vvo[i] = (unsigned)(r * p_vrowb + ((lo5 ^ r) << 4));
        }
    };
    setup_p();
    auto produce = [&](auto stc) {
        constexpr int PS = decltype(stc)::value;
        if (!p_live) return;
        unsigned k0 = kvo[0], k1 = kvo[1], v0 = vvo[0], v1 = vvo[1];
        if (p_kt + 16 > p_it.Sk) {  // the unit's last tile: rows past its end arrive as zeros
            const int left = p_it.Sk - p_kt - 4 * wave;
            if (hi >= left) { k0 = OOB; v0 = OOB; }
            if (2 + hi >= left) { k1 = OOB; v1 = OOB; }
        }
        char* const st = lds + PS * RSTAGE + wave * 2048;
        if (!(ABL & 2)) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lptr_t)st, 16, k0, p_kt * p_krowb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lptr_t)(st + 1024), 16, k1, p_kt * p_krowb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lptr_t)(st + RKV), 16, v0, p_kt * p_vrowb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lptr_t)(st + RKV + 1024), 16, v1, p_kt * p_vrowb, 0, 0);
            issued += 4;
        }
        mark[PS] = issued;
        p_kt += 16;
        if (p_kt >= p_it.Sk) {
            p_idx += stride;
            p_live = next_valid(p_idx, p_it);
            p_kt = 0;
            if (p_live) setup_p();
        }
    };

    // ---- fragment addresses (bytes from the start of LDS).  Slot of chunk ch in row r: ch ^ (r & 15).
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
    unsigned ka[4], qa[4], va[4][4];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) {
        // K / q: row c16, chunk g4 + 4 c: slot bits [1:0] = g4 ^ (c16 & 3), [3:2] = (c & 3) ^ (c16 >> 2), [4] = c >> 2 (an immediate)
        ka[cl] = lds_base + (unsigned)(c16 * RROWB + ((((cl ^ (c16 >> 2)) << 2) | (g4 ^ (c16 & 3))) << 4));
        qa[cl] = ka[cl] + (unsigned)(wave * RQ_WAVE);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            // V: row 4 g4 + r, element c16 + 16 c: chunk (c16 >> 2) + 4 c: slot bits [1:0] = (c16 >> 2) ^ r, [3:2] = (c & 3) ^ g4
            va[r][cl] = lds_base + (unsigned)((4 * g4 + r) * RROWB + ((((cl ^ g4) << 2) | ((c16 >> 2) ^ r)) << 4) + (c16 & 3) * 4);
    }
#define RING_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define RING_RD2(dst, addr, o0) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(o0), "n"((o0) + 1) : "memory")

    // ---- consumer state
    Item cur = p_it, nxt = p_it;
    int c_idx = p_idx, n_idx = p_idx;
    bool have_next = false;
    int c_kt = 0;
    f32x4 qf[8], oacc[8];
    float m_run = -INFINITY, l_run = 0.f;

    issue_q(cur);
    produce(std::integral_constant<int, 0>{});
    produce(std::integral_constant<int, 1>{});

    auto tile = [&](auto stc) -> bool {
        constexpr int ST = decltype(stc)::value;
        if (c_kt == 0) {  // a work item begins: its queries have been on their way since the previous item began
            wait_vm(issued - mark_q);
            if constexpr (!(ABL & 1)) {
#pragma unroll
                for (int c = 0; c < 8; ++c) RING_RD128(qf[c], qa[c & 3], RQ_OFF + (c >> 2) * 256);
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(qf[4]), "+v"(qf[5]), "+v"(qf[6]), "+v"(qf[7])::"memory");
            }
            n_idx = c_idx + stride;
            have_next = next_valid(n_idx, nxt);
            if (have_next) issue_q(nxt);
#pragma unroll
            for (int c = 0; c < 8; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            m_run = -INFINITY;
            l_run = 0.f;
        }
        wait_vm(issued - mark[ST]);
        __builtin_amdgcn_s_barrier();  // every wave's share of this tile has landed; every wave has left the previous tile's stage
        produce(std::integral_constant<int, (ST + 2) % RNST>{});

        if constexpr (!(ABL & 1)) {
        f32x4 kf[8];
        f32x2 vf[4][4];
#pragma unroll
        for (int c = 0; c < 8; ++c) RING_RD128(kf[c], ka[c & 3], ST * RSTAGE + (c >> 2) * 256);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) RING_RD2(vf[r][cl], va[r][cl], (ST * RSTAGE + RKV) / 256);
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7])::"memory");
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c & 1) {
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][0], qf[c][0], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][1], qf[c][1], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][2], qf[c][2], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][3], qf[c][3], a1, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][0], qf[c][0], a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][1], qf[c][1], a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][2], qf[c][2], a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][3], qf[c][3], a0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 2; r < 4; ++r)
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) RING_RD2(vf[r][cl], va[r][cl], (ST * RSTAGE + RKV) / 256);
        f32x4 sc;
        const int key0 = c_kt + 4 * g4;
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[r] = (key0 + r < cur.Sk) ? (a0[r] + a1[r]) * a.scale : -INFINITY;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[r]);
        mx = max_xor32(max_xor16(mx));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = __expf(sc[r] - m_new);
            rs += sc[r];
        }
        rs = sum_xor32(sum_xor16(rs));
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int c = 0; c < 8; ++c) oacc[c] *= alpha;
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[0][2]), "+v"(vf[0][3]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[1][2]), "+v"(vf[1][3]),
                       "+v"(vf[2][0]), "+v"(vf[2][1]), "+v"(vf[2][2]), "+v"(vf[2][3]), "+v"(vf[3][0]), "+v"(vf[3][1]), "+v"(vf[3][2]), "+v"(vf[3][3])::"memory");
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][c & 3][c >> 2], sc[r], oacc[c], 0, 0, 0);

        }
        c_kt += 16;
        if (c_kt < cur.Sk) return false;
        // ---- the work item ends: normalise and store this wave's 16 rows (attn_fwd_f32_simple_kernel's epilogue)
        const int qi = cur.qb * 64 + wave * 16 + c16;
        if (cur.qb * 64 + wave * 16 < cur.Sq) {  // wave-uniform: the stores are issued (and counted) or not
            if (qi < cur.Sq) {
                const float inv = 1.f / l_run;
                float* op = a.o + (cur.q0 + (long long)qi * cur.q_rs) * a.ldo + cur.h * RDH;
                if (!a.o_sp16) {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        *reinterpret_cast<float4*>(op + 4 * g4 + c * 16) = make_float4(oacc[c][0] * inv, oacc[c][1] * inv, oacc[c][2] * inv, oacc[c][3] * inv);
                } else {
                    float m = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        half4v h4, l4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = oacc[c][j] * inv;
                            _Float16 h1, l1;
                            split_f16(v, h1, l1);
                            h4[j] = h1; l4[j] = l1;
                            m = fmaxf(m, fabsf(v));
                        }
                        char* dst = reinterpret_cast<char*>(op + c * 16 + 8 * (g4 >> 1)) + 8 * (g4 & 1);
                        *reinterpret_cast<half4v*>(dst) = h4;
                        *reinterpret_cast<half4v*>(dst + 16) = l4;
                    }
                    if (a.guard && !(m < 65000.f)) atomicOr(a.guard, 1);
                }
            }
            issued += a.o_sp16 ? 16 : 8;
        }
        if (!have_next) return true;
        cur = nxt;
        c_idx = n_idx;
        c_kt = 0;
        return false;
    };
    for (;;) {
        if (tile(std::integral_constant<int, 0>{})) break;
        if (tile(std::integral_constant<int, 1>{})) break;
        if (tile(std::integral_constant<int, 2>{})) break;
    }
}

}  // namespace

template <int ABL>
static int launch_ring(const AttnRArgs& a, int grid, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        SOLA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_f32_ring_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, RLDS));
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_fwd_f32_ring_kernel<ABL>, dim3((unsigned)grid), dim3(256), RLDS, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int g_attn_ring_blocks = 2;  // sola_tune "attn_ring_blocks": persistent blocks per CU (80 KB of LDS each: two fit)
int g_attn_ring_remap = 1;   // sola_tune "attn_ring_remap": consecutive work items on one XCD
int g_attn_ring_ablate = 0;  // sola_tune "attn_ring_ablate"
void sola_attn_set_ring_blocks(int v) { g_attn_ring_blocks = v < 1 ? 1 : v; }
void sola_attn_set_ring_remap(int v) { g_attn_ring_remap = v; }
void sola_attn_set_ring_ablate(int v) { g_attn_ring_ablate = v; }
int g_attn_ring = 0;  // sola_tune "attn_ring": 1 = this shape where attn_simple.hip's applies, 2 = also in place of attn_res.hip's (A/B)
void sola_attn_set_ring(int v) { g_attn_ring = v; }

// f32 q / k / v, head_dim 128, inference (no log-sum-exp, no dropout), more than 16 keys or queries; the byte offsets inside a unit fit
// the buffer instructions' 32-bit offsets
bool attention_ring_supported(const AttnDesc& d) {
    if (!g_attn_ring || d.lse || d.drop.enabled || d.in_sp16 || d.split_math || d.DH != 128 || d.k_private > 0) return false;
    if (!(d.Sq > 16 || d.Sk > 16)) return false;
    const long long k_rs = d.q_units ? 64 : d.k_rs, q_rs = d.q_units ? 64 : d.q_rs;  // unit tables: row strides are at most T' (<= 64)
    const long long kb = ((long long)d.Sk + 16) * k_rs * std::max(d.ldk, d.ldv) * 4, qb = 80 * q_rs * d.ldq * 4;
    return kb < (1ll << 31) && qb < (1ll << 31);
}

int launch_attention_ring(const AttnDesc& d, hipStream_t s) {
    AttnRArgs a;
    a.q = d.q; a.k = d.k; a.v = d.v; a.o = d.o;
    a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo;
    a.G = d.G; a.H = d.H; a.Sq = d.Sq; a.Sk = d.Sk; a.inner = d.inner;
    a.nqb = (d.Sq + 63) / 64;
    a.q_outer = d.q_outer; a.q_inner = d.q_inner; a.q_rs = d.q_rs;
    a.k_outer = d.k_outer; a.k_inner = d.k_inner; a.k_rs = d.k_rs;
    a.scale = d.scale; a.o_sp16 = d.o_sp16; a.guard = d.o_sp16 ? d.guard : nullptr;
    a.q_units = d.q_units; a.k_units = d.q_units ? (d.k_units ? d.k_units : d.q_units) : nullptr;
    const long long items = (long long)d.G * d.H * a.nqb;
    SOLA_ARG(items < (1ll << 31), "attention: grid too large");
    a.n_items = (int)items;
    const double elems = (double)d.G * d.H * d.DH;
    SolaProfScope prof(SOLA_PROF_ATTN, s, 4.0 * elems * d.Sq * d.Sk, 4.0 * elems * (2.0 * d.Sq + 2.0 * d.Sk));
    const long long want = (long long)g_attn_ring_blocks * sola_cu_count();
    const int grid = (int)std::min(want, items);
    a.xcd_remap = (grid % 8 == 0 && g_attn_ring_remap) ? 1 : 0;
    if (g_attn_ring_ablate == 1) return launch_ring<1>(a, grid, s);
    if (g_attn_ring_ablate == 2) return launch_ring<2>(a, grid, s);
    if (g_attn_ring_ablate == 3) return launch_ring<3>(a, grid, s);
    return launch_ring<0>(a, grid, s);
}
#endif
