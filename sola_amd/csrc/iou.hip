// Pairwise mask IoU for the track de-duplication predicate (track_generation/seg_utils.py:128-142 compute_mask_iou,
// :109-125 compute_masklet_iou; call sites generate_tokens_grid.py:266-278, generate_tokens_gdino.py:288-300).
//
// The reference evaluates sum(A*B) and sum(A+B) on float {0,1} images, one pair per call, with two host syncs per
// pair.  Those sums are exact integers (< 2^24), so the predicate is integer popcount arithmetic:
//   1. mask_pack : every mask is read ONCE from HBM (the only large traffic: (P+R)*H*W bytes), thresholded (!= 0),
//                  optionally resampled with ATen's nearest rule (generate_tokens_grid.py:271-272), and written as
//                  1 bit/pixel together with its area (popcount).  Pure HBM-bound byte streaming.
//   2. mask_pair : inter[p,r] = popcount(Abits & Bbits) over 1/8-size L2-resident bit rows; union = |A|+|B|-inter.
// Counts are int64 and order-independent (integer adds), so results are bit-exact and deterministic.
#include <algorithm>

#include "kernels.h"

namespace {

struct PackArgs {
    const void* src;
    uint32_t* bits;
    unsigned long long* area;
    int h, w, H, W;
    long long hw_src, HW, words;
    float sy, sx;
    int identity;
    // mask_pack_u8_stream_kernel only: masks n >= n_split come from the SECOND set (src2 -> bits2 / area2, index n - n_split):
    // both mask sets of a P x R matrix in one launch (launch_mask_pack_pair)
    int n_split;
    const void* src2;
    uint32_t* bits2;
    unsigned long long* area2;
};

template <typename T>
__device__ __forceinline__ bool is_set(T v) { return v != (T)0; }

// One lane packs 16 consecutive destination pixels; lane pairs are OR-combined into one 32-bit word.
template <typename T>
__global__ __launch_bounds__(256) void mask_pack_kernel(const PackArgs a) {
    __shared__ int red[4];
    const int n = blockIdx.y;
    const long long run = (long long)blockIdx.x * 256 + threadIdx.x;  // 16-pixel run index inside this mask
    const long long p0 = run * 16;
    const T* src = reinterpret_cast<const T*>(a.src) + (long long)n * a.hw_src;
    unsigned bits = 0;
    if (p0 < a.HW) {
        if (a.identity && p0 + 16 <= a.HW && (a.hw_src * (long long)sizeof(T)) % 16 == 0 &&
            (reinterpret_cast<uintptr_t>(a.src) & 15) == 0) {
            if constexpr (sizeof(T) == 1) {
                const uint4 v = *reinterpret_cast<const uint4*>(src + p0);
                const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // word-parallel "byte != 0" -> one bit per byte: high bit of each non-zero byte, moved to bit 0 of its
                    // byte, then the four byte flags are gathered into a nibble by one multiply (no carries reach bits 24-27)
                    const unsigned w = wv[j];
                    const unsigned nz = ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u) >> 7;
                    bits |= ((nz * 0x01020408u) >> 24 & 0xfu) << (4 * j);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 v = *reinterpret_cast<const float4*>(src + p0 + 4 * j);
                    bits |= (v.x != 0.f ? 1u : 0u) << (4 * j) | (v.y != 0.f ? 1u : 0u) << (4 * j + 1) |
                            (v.z != 0.f ? 1u : 0u) << (4 * j + 2) | (v.w != 0.f ? 1u : 0u) << (4 * j + 3);
                }
            }
        } else {
            int y = (int)(p0 / a.W);
            int x = (int)(p0 - (long long)y * a.W);
            for (int i = 0; i < 16; ++i) {
                if (p0 + i >= a.HW) break;
                int sy = y, sx = x;
                if (!a.identity) {
                    // ATen nearest: min(floor(dst * float(in/out)), in - 1), product in fp32
                    sy = a.h == a.H ? y : min((int)floorf((float)y * a.sy), a.h - 1);
                    sx = a.w == a.W ? x : min((int)floorf((float)x * a.sx), a.w - 1);
                }
                bits |= (is_set(src[(long long)sy * a.w + sx]) ? 1u : 0u) << i;
                if (++x == a.W) { x = 0; ++y; }
            }
        }
    }
    const unsigned other = __shfl_xor(bits, 1, 64);
    if ((threadIdx.x & 1) == 0) {
        const long long word = run >> 1;
        if (word < a.words) a.bits[(long long)n * a.words + word] = bits | (other << 16);
    }
    int cnt = __popc(bits);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = red[0] + red[1] + red[2] + red[3];
        if (tot) atomicAdd(&a.area[n], (unsigned long long)tot);
    }
}

// Nearest-resample pack (generate_tokens_grid.py:271-272: prompt masks of another resolution): the generic kernel above
// walks 16 destination pixels per lane with a source-index computation and a scattered byte load for each (17-24 % of the
// HBM peak at 720x1280 / 1080x1920 -> 540x960).  Here a block owns RPB destination rows of one mask:
//   1. their source rows (only the rows ATen's rule selects are touched at all) are read ONCE with 16-byte loads, several in
//      flight per lane, thresholded and written to LDS as BITS (40 words for a 1280-pixel row);
//   2. a lane builds one destination word from the (at most four) consecutive source words its 32 pixels fall into: four
//      LDS reads per word instead of one per pixel (the per-pixel byte gather from LDS was latency-bound at the same 160 us
//      as the gather from HBM), the column rule and the bit select run in registers.
// Needs W % 32 == 0 and a horizontal scale of at most 3 (a word's 32 pixels then span < 128 source pixels).
constexpr int RS_RPB = 8, RS_MAXW = 4096, RS_WORDS = RS_MAXW / 32 + 4;
__device__ __forceinline__ unsigned pack16(const uint4 v);
template <typename T>
__global__ __launch_bounds__(256) void mask_pack_resample_kernel(const PackArgs a) {
    __shared__ unsigned sbits[RS_RPB][RS_WORDS];
    __shared__ int red[4];
    const int n = blockIdx.y;
    const T* src = reinterpret_cast<const T*>(a.src) + (long long)n * a.hw_src;
    int cnt = 0;
    // A block walks row chunks blockIdx.x, + gridDim.x, ... and adds its pixel count to the mask's area ONCE: one 64-bit
    // atomic per 8-row chunk (17 K of them at 256 masks, eight masks to a cache line) serialised in L2 and was the whole
    // 160 us of this kernel - and of the per-pixel kernel before it.
    for (int y0 = blockIdx.x * RS_RPB; y0 < a.H; y0 += gridDim.x * RS_RPB) {
    if (y0 != (int)blockIdx.x * RS_RPB) __syncthreads();  // the previous chunk's LDS rows have been consumed
    const int nrows = min(RS_RPB, a.H - y0);
    const int sw = (a.w + 31) >> 5;  // source words per row
    // rows made of whole, 16-byte aligned 32-pixel groups: vector loads
    const bool vec = (a.w & 31) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && ((a.hw_src * (long long)sizeof(T)) & 15) == 0;
    const int total = nrows * sw;
    for (int e0 = threadIdx.x; e0 < total; e0 += 2 * 256) {
        unsigned word[2] = {0u, 0u};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = e0 + q * 256;
            if (e >= total) continue;
            const int r = e / sw, jw = e - r * sw;
            const int y = y0 + r;
            const int sy = a.h == a.H ? y : min((int)floorf((float)y * a.sy), a.h - 1);  // ATen nearest, fp32 product
            const T* p = src + (long long)sy * a.w + jw * 32;
            if (vec) {
                if constexpr (sizeof(T) == 1) {
                    const uint4 lo = reinterpret_cast<const uint4*>(p)[0], hi = reinterpret_cast<const uint4*>(p)[1];
                    word[q] = pack16(lo) | (pack16(hi) << 16);
                } else {
                    unsigned wv = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float4 v = reinterpret_cast<const float4*>(p)[k];
                        wv |= ((v.x != 0.f ? 1u : 0u) | (v.y != 0.f ? 2u : 0u) | (v.z != 0.f ? 4u : 0u) | (v.w != 0.f ? 8u : 0u)) << (4 * k);
                    }
                    word[q] = wv;
                }
            } else {
                unsigned wv = 0;
                const int lim = min(32, a.w - jw * 32);
                for (int k = 0; k < lim; ++k) wv |= (is_set(p[k]) ? 1u : 0u) << k;
                word[q] = wv;
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = e0 + q * 256;
            if (e < total) {
                const int r = e / sw;
                sbits[r][e - r * sw] = word[q];
            }
        }
    }
    if (threadIdx.x < RS_RPB * 4) sbits[threadIdx.x >> 2][sw + (threadIdx.x & 3)] = 0u;  // the window may run past the row's last word
    __syncthreads();
    const int wpr = a.W >> 5;
    for (int idx = threadIdx.x; idx < nrows * wpr; idx += 256) {
        const int r = idx / wpr, j = idx - r * wpr;
        const int x0 = 32 * j;
        const int s0 = a.w == a.W ? x0 : min((int)floorf((float)x0 * a.sx), a.w - 1);
        const int wb = s0 >> 5;
        const unsigned w0 = sbits[r][wb], w1 = sbits[r][wb + 1], w2 = sbits[r][wb + 2], w3 = sbits[r][wb + 3];
        unsigned bits = 0;
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const int x = x0 + b;
            const int sx = a.w == a.W ? x : min((int)floorf((float)x * a.sx), a.w - 1);
            const int k = sx - (wb << 5);  // 0 .. 127
            const unsigned t0 = (k & 32) ? w1 : w0, t1 = (k & 32) ? w3 : w2;
            const unsigned t = (k & 64) ? t1 : t0;
            bits |= ((t >> (k & 31)) & 1u) << b;
        }
        a.bits[(long long)n * a.words + (long long)(y0 + r) * wpr + j] = bits;
        cnt += __popc(bits);
    }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = red[0] + red[1] + red[2] + red[3];
        if (tot) atomicAdd(&a.area[n], (unsigned long long)tot);
    }
}

// Streaming fast path for the common case (uint8 masks already at the comparison resolution, 32-byte aligned rows of
// H*W % 32 == 0 pixels): a lane packs one whole 32-bit word from 32 bytes, four words per lane with all eight 16-byte
// loads in flight before the first use, one area atomic per 32 KiB of mask.
__device__ __forceinline__ unsigned pack16(const uint4 v) {
    const unsigned wv[4] = {v.x, v.y, v.z, v.w};
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned w = wv[j];
        const unsigned nz = ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u) >> 7;
        bits |= ((nz * 0x01020408u) >> 24 & 0xfu) << (4 * j);
    }
    return bits;
}

__global__ __launch_bounds__(256) void mask_pack_u8_stream_kernel(const PackArgs a) {
    __shared__ int red[4];
    constexpr int IT = 4;
    const bool second = (int)blockIdx.y >= a.n_split;
    const int n = second ? (int)blockIdx.y - a.n_split : (int)blockIdx.y;
    uint32_t* const bits = second ? a.bits2 : a.bits;
    unsigned long long* const area = second ? a.area2 : a.area;
    const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(second ? a.src2 : a.src) + (long long)n * a.hw_src);
    const long long w0 = (long long)blockIdx.x * (256 * IT) + threadIdx.x;
    uint4 lo[IT], hi[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const long long w = w0 + it * 256;
        const bool ok = w < a.words;
        lo[it] = ok ? src[2 * w] : make_uint4(0u, 0u, 0u, 0u);
        hi[it] = ok ? src[2 * w + 1] : make_uint4(0u, 0u, 0u, 0u);
    }
    int cnt = 0;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const long long w = w0 + it * 256;
        const unsigned word = pack16(lo[it]) | (pack16(hi[it]) << 16);
        if (w < a.words) bits[(long long)n * a.words + w] = word;
        cnt += __popc(word);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = red[0] + red[1] + red[2] + red[3];
        if (tot) atomicAdd(&area[n], (unsigned long long)tot);
    }
}

struct PairArgs {
    const uint32_t *a_bits, *b_bits;
    const long long *a_area, *b_area;
    const int32_t* a_frame;
    long long words;
    int P, T, R;
    long long *inter, *uni;
};

constexpr int PB = 4;  // A masks per block (share the B words)

__global__ __launch_bounds__(256) void mask_pair_kernel(const PairArgs a) {
    __shared__ int red[4][PB];
    const int r = blockIdx.x;
    const int pbase = blockIdx.y * PB;
    const int frame = a.a_frame ? a.a_frame[r] : 0;
    const uint32_t* bw = a.b_bits + (long long)r * a.words;
    const uint32_t* aw[PB];
    int np = 0;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int p = min(pbase + j, a.P - 1);
        aw[j] = a.a_bits + ((long long)p * a.T + frame) * a.words;
        if (pbase + j < a.P) np = j + 1;
    }
    int cnt[PB] = {0, 0, 0, 0};
    const long long w4 = a.words >> 2;
    const bool vec_ok = (a.words & 3) == 0;  // rows stay 16-byte aligned
    if (vec_ok) {
        for (long long i = threadIdx.x; i < w4; i += 256) {
            const uint4 b = reinterpret_cast<const uint4*>(bw)[i];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const uint4 v = reinterpret_cast<const uint4*>(aw[j])[i];
                cnt[j] += __popc(v.x & b.x) + __popc(v.y & b.y) + __popc(v.z & b.z) + __popc(v.w & b.w);
            }
        }
    } else {
        for (long long i = threadIdx.x; i < a.words; i += 256) {
            const uint32_t b = bw[i];
#pragma unroll
            for (int j = 0; j < PB; ++j) cnt[j] += __popc(aw[j][i] & b);
        }
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        int c = cnt[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = c;
    }
    __syncthreads();
    if (threadIdx.x < np) {
        const int j = threadIdx.x, p = pbase + j;
        const long long in = (long long)red[0][j] + red[1][j] + red[2][j] + red[3][j];
        a.inter[(long long)p * a.R + r] = in;
        a.uni[(long long)p * a.R + r] = a.a_area[(long long)p * a.T + frame] + a.b_area[r] - in;  // sum(A+B) - inter
    }
}

// ---- one-launch path for the de-dup loop's real call sizes (generate_tokens_grid.py:266-278: P <= 4 new tracks against
//      16-256 prompts, uint8 masks already at the comparison resolution) --------------------------------------------------
// Round 6: ONE kernel and nothing else on the stream - no memset, no atomically accumulated table.
//   * block (chunk, prompt group): 512 words (16 KiB of each mask) x G prompts; every 16-byte load of the block - the P
//     track slices and the G prompt slices - is issued before the first use, each mask byte leaves HBM once (the tracks
//     again per prompt group, from L2);
//   * bits are packed in registers (never written), popc(A & B), |B|, |A| summed over the wave with DPP adds + two
//     permlane swaps (integer sums: exact in any order), over the four waves through LDS, and the block's 5 G + 4 counts
//     are STORED to its own row of a partial table;
//   * the last block of a prompt group to finish (a ticket per group) adds the group's rows in chunk order and writes
//     inter / union.  The tickets live in the library (g_iou_ticket: zero when the code object is loaded, put back to zero
//     by the block that takes the last one), handed out as a ring range per launch - so a call leaves no state behind and
//     needs no zeroed memory in front of it.  4096 tickets: launches on different streams overlap safely as long as fewer
//     than 4096 prompt groups are in flight at once.
// The grid is sized by the bytes read: 1 / 2 / 4 / 8 prompts per block for R <= 16 / <= 32 / <= 128 / more, two slices in flight per
// batch (512-1024 blocks at 540 x 960; onepass_shape).
struct OnepassArgs {
    const uint8_t *a, *b;
    long long hw, words;
    int P, R, chunks, groups, nb;
    unsigned ticket_base;
    unsigned acc_base;  // PK: first line of this call's pair words in g_iou_acc (pair (prompt r, track p) at acc_base + 4 r + p)
    unsigned* part;  // [groups][chunks][5 G + 4]: per prompt j of the group inter[p] (p < 4), |B_j|; then |A_p| (p < 4)
    long long *inter, *uni;
};
constexpr int FUSED_MAXP = 4;
constexpr int OP_WPT = 2;               // 32-pixel words per thread
constexpr int OP_CHUNK = 256 * OP_WPT;  // words per block
constexpr int OP_TICKETS = 4096;
constexpr int OP_MAXV = 64;             // counts per block: 5 per prompt (at most 12 prompts) + 4
constexpr int OP_MAX_CHUNKS = 128;      // the group's fold walks this many rows at most (masks of up to 2 M pixels)
constexpr int OP_TICKET_PITCH = 32;    // a ticket per 128-byte line: agent-scope atomics on ONE line are serialised where they execute (~50 ns each:
                                       // 512 blocks on 16 neighbouring words took 25 us); a line per prompt group leaves 32 of them in a row
__device__ unsigned g_iou_ticket[OP_TICKETS * OP_TICKET_PITCH];
// PK (round 6): masks of fewer than 2^19 pixels (540 x 960 = 518 400) - one 64-bit word per (track, prompt) pair holds
// [arrivals:7 | |A_p|:19 | |B_r|:19 | inter:19]; every block ADDS its partial counts and a 1 to it with an agent-scope atomic that returns the
// old value, and the lane that sees chunks - 1 arrivals owns the totals: it writes inter / union and puts the word back to zero.  One memory
// round trip behind the loads where the ticket form has three (row store, ticket, fold loads).  A word per 128-byte line, as the tickets.
constexpr int OP_ACC_LINES = 16384;  // ring of pair words (2 MB); a call takes 4 R of them
constexpr int OP_ACC_PITCH = 16;     // 64-bit words per line
__device__ unsigned long long g_iou_acc[(size_t)OP_ACC_LINES * OP_ACC_PITCH];

// Integer sum over the 64 lanes, every lane gets it: DPP adds inside the 16-lane rows, v_permlane16_swap across the row pairs,
// v_permlane32_swap across the halves (common.h: half_sum32 / wave_sum_dpp, on integers)
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
    unsigned x = v, y = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    v = x + y;
    x = v; y = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}

template <int G, bool PK = false>
__global__ __launch_bounds__(256) void mask_iou_onepass_kernel(const OnepassArgs a) {
    __shared__ unsigned red[4][OP_MAXV];
    __shared__ unsigned tot[4][OP_MAXV];
    __shared__ int is_last;
    const int chunk = blockIdx.x, grp = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int GB = G * a.nb;         // prompts per block: nb batches of G (the G slices of a batch are all in flight at once)
    const int NV = 5 * GB + 4;       // counts per block: per prompt inter[p] (p < 4) and |B|, then |A_p|
    const int r0 = grp * GB;
    // every load is unconditional (a predicated 16-byte load compiles to four predicated dword loads: 160 of them per thread made this
    // kernel 2.5x slower): out-of-range words / absent tracks read a clamped address and their PACKED word is zeroed below
    bool ok[OP_WPT];
    long long wc[OP_WPT];
    unsigned wa[OP_WPT][FUSED_MAXP];
    {
        uint4 la[FUSED_MAXP][OP_WPT][2];
#pragma unroll
        for (int it = 0; it < OP_WPT; ++it) {
            const long long w = (long long)chunk * OP_CHUNK + it * 256 + threadIdx.x;
            ok[it] = w < a.words;
            wc[it] = ok[it] ? w : a.words - 1;
#pragma unroll
            for (int p = 0; p < FUSED_MAXP; ++p) {
                const uint4* src = reinterpret_cast<const uint4*>(a.a + (long long)min(p, a.P - 1) * a.hw) + 2 * wc[it];
                la[p][it][0] = src[0];
                la[p][it][1] = src[1];
            }
        }
        uint4 lb[G][OP_WPT][2];
        auto load_b = [&](int bt) {
#pragma unroll
            for (int it = 0; it < OP_WPT; ++it)
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const uint4* src = reinterpret_cast<const uint4*>(a.b + (long long)min(r0 + bt * G + j, a.R - 1) * a.hw) + 2 * wc[it];
                    lb[j][it][0] = src[0];
                    lb[j][it][1] = src[1];
                }
        };
        load_b(0);
        // two 16-bit fields per word: a thread sees 64 pixels per mask, a wave 4096
        unsigned ca[2] = {0u, 0u};
#pragma unroll
        for (int it = 0; it < OP_WPT; ++it) {
#pragma unroll
            for (int p = 0; p < FUSED_MAXP; ++p) wa[it][p] = (ok[it] && p < a.P) ? (pack16(la[p][it][0]) | (pack16(la[p][it][1]) << 16)) : 0u;
            ca[0] += __popc(wa[it][0]) | (__popc(wa[it][1]) << 16);
            ca[1] += __popc(wa[it][2]) | (__popc(wa[it][3]) << 16);
        }
        {
            const unsigned a01 = wave_sum_u32(ca[0]), a23 = wave_sum_u32(ca[1]);
            if (lane == 0) {
                red[wave][5 * GB + 0] = a01 & 0xffffu; red[wave][5 * GB + 1] = a01 >> 16;
                red[wave][5 * GB + 2] = a23 & 0xffffu; red[wave][5 * GB + 3] = a23 >> 16;
            }
        }
        for (int bt = 0; bt < a.nb; ++bt) {
            unsigned wb[G][OP_WPT];
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int it = 0; it < OP_WPT; ++it) wb[j][it] = ok[it] ? (pack16(lb[j][it][0]) | (pack16(lb[j][it][1]) << 16)) : 0u;
            if (bt + 1 < a.nb) load_b(bt + 1);  // the next batch's slices travel under this batch's counts
#pragma unroll
            for (int j = 0; j < G; ++j) {
                unsigned c01 = 0u, c23 = 0u, cb = 0u;
#pragma unroll
                for (int it = 0; it < OP_WPT; ++it) {
                    c01 += __popc(wa[it][0] & wb[j][it]) | (__popc(wa[it][1] & wb[j][it]) << 16);
                    c23 += __popc(wa[it][2] & wb[j][it]) | (__popc(wa[it][3] & wb[j][it]) << 16);
                    cb += __popc(wb[j][it]);
                }
                const unsigned s01 = wave_sum_u32(c01), s23 = wave_sum_u32(c23), sb = wave_sum_u32(cb);
                if (lane == 0) {
                    unsigned* r = &red[wave][5 * (bt * G + j)];
                    r[0] = s01 & 0xffffu; r[1] = s01 >> 16; r[2] = s23 & 0xffffu; r[3] = s23 >> 16; r[4] = sb;
                }
            }
        }
    }
    __syncthreads();
    if constexpr (PK) {
        if ((int)threadIdx.x < GB * FUSED_MAXP) {
            const int j = threadIdx.x / FUSED_MAXP, p = threadIdx.x - j * FUSED_MAXP;
            const int r = r0 + j;
            if (r < a.R && p < a.P) {
                const unsigned long long in = (red[0][5 * j + p] + red[1][5 * j + p]) + (red[2][5 * j + p] + red[3][5 * j + p]);
                const unsigned long long nb_ = (red[0][5 * j + 4] + red[1][5 * j + 4]) + (red[2][5 * j + 4] + red[3][5 * j + 4]);
                const unsigned long long na = (red[0][5 * GB + p] + red[1][5 * GB + p]) + (red[2][5 * GB + p] + red[3][5 * GB + p]);
                const unsigned long long mine = in | (nb_ << 19) | (na << 38) | (1ull << 57);
                unsigned long long* w = &g_iou_acc[(size_t)((a.acc_base + 4u * (unsigned)r + (unsigned)p) & (OP_ACC_LINES - 1)) * OP_ACC_PITCH];
                const unsigned long long old = __hip_atomic_fetch_add(w, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(old >> 57) == a.chunks - 1) {  // every other block of this group has added its counts
                    const unsigned long long t = old + mine;
                    const long long ti = (long long)(t & 0x7ffffull), tb = (long long)((t >> 19) & 0x7ffffull), ta = (long long)((t >> 38) & 0x7ffffull);
                    a.inter[(long long)p * a.R + r] = ti;
                    a.uni[(long long)p * a.R + r] = ta + tb - ti;  // sum(A + B) - inter (seg_utils.py:133-134)
                    __hip_atomic_store(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        return;
    }
    unsigned* const rows = a.part + (long long)grp * a.chunks * NV;
    // ---- the group's last block folds.  No agent-scope FENCE: on gfx950 a __threadfence() is an L2 write-back + invalidate, and one per
    //      block (512-2048 of them) serialised per XCD made this kernel 3-8x slower than its predecessor (51 / 120 / 285 us at R = 16 / 64 /
    //      256, profiles/r06_iou.txt).  The row is written with agent-scope ATOMIC stores (performed at the level all XCDs share), the
    //      storing lanes wait for their completion (workgroup-scope release = s_waitcnt), then the ticket; the folding block reads the
    //      rows with agent-scope atomic loads behind the ticket it took.
    if ((int)threadIdx.x < NV)
        __hip_atomic_store(&rows[(long long)chunk * NV + threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    const unsigned slot = ((a.ticket_base + (unsigned)grp) & (OP_TICKETS - 1)) * OP_TICKET_PITCH;
    if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(&g_iou_ticket[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.chunks - 1);
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    {   // four stripes of chunks x 64 value slots
        const int v = threadIdx.x & 63, stripe = threadIdx.x >> 6;
        unsigned sum = 0;
        if (v < NV)
            for (int c = stripe; c < a.chunks; c += 4) sum += __hip_atomic_load(&rows[(long long)c * NV + v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot[stripe][v] = sum;
    }
    if (threadIdx.x == 0) __hip_atomic_store(&g_iou_ticket[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if ((int)threadIdx.x < GB * FUSED_MAXP) {
        const int j = threadIdx.x / FUSED_MAXP, p = threadIdx.x - j * FUSED_MAXP;
        const int r = r0 + j;
        if (r < a.R && p < a.P) {
            long long in = 0, ab = 0, aa = 0;
#pragma unroll
            for (int st = 0; st < 4; ++st) { in += tot[st][5 * j + p]; ab += tot[st][5 * j + 4]; aa += tot[st][5 * GB + p]; }
            a.inter[(long long)p * a.R + r] = in;
            a.uni[(long long)p * a.R + r] = aa + ab - in;  // sum(A + B) - inter (seg_utils.py:133-134)
        }
    }
}

}  // namespace

int g_iou_shape = 0;  // sola_tune "iou_shape": 0 = by R; else 16 * G + nb (G in {1, 2, 4} slices in flight, nb batches per block; G * nb <= 12)
void sola_iou_set_shape(int v) { g_iou_shape = v; }
static void onepass_shape(int R, int* G, int* nb) {
    if (g_iou_shape > 0 && (g_iou_shape >> 4) * (g_iou_shape & 15) <= 12 && (g_iou_shape & 15) > 0 &&
        ((g_iou_shape >> 4) == 1 || (g_iou_shape >> 4) == 2 || (g_iou_shape >> 4) == 4)) {
        *G = g_iou_shape >> 4; *nb = g_iou_shape & 15;
        return;
    }
    // measured (tools/iou_sweep.py, P = 4, 540 x 960; kernel us at (G, nb)): R = 16: 10.5-10.9 for every one- or two-prompt shape (the launch's
    // latency chain: loads, row store + ticket, fold); R = 64: 20.4 (1,1) 14.8 (2,1) 13.5 (2,2); R = 256: 56.0 (1,1) 32.0 (4,1) 29.3 (2,2) 27.8 (2,4)
    *G = R <= 16 ? 1 : 2;
    *nb = R <= 32 ? 1 : (R <= 128 ? 2 : 4);
}
size_t mask_iou_fused_scratch_bytes(int P, int R, long long words) {
    (void)P;
    // an upper bound over every block shape: ceil(R / GB) groups x (5 GB + 4) counts <= 9 R + 64 for 1 <= GB <= 12
    const size_t chunks = (size_t)((words + OP_CHUNK - 1) / OP_CHUNK);
    return ((size_t)9 * R + 64) * chunks * sizeof(unsigned);
}

int g_iou_packed = 1;  // sola_tune "iou_packed": 0 = the ticket form for every mask size (A/B, tests)
void sola_iou_set_packed(int v) { g_iou_packed = v; }
int g_iou_fused = 1;  // sola_tune "iou_fused": 0 forces the pack + pair path (A/B, tests)
void sola_iou_set_fused(int v) { g_iou_fused = v; }

// true if the call was served by the one-launch kernel
bool launch_mask_iou_fused(const void* am, const void* bm, int elem_type, int P, int R, int H, int W, int h, int w, long long* inter,
                           long long* uni, void* scratch, size_t scratch_bytes, hipStream_t s, int* status) {
    *status = SOLA_OK;
    const long long hw = (long long)H * W;
    // Round 3 measured the earlier fused kernel (atomics + memset, 16 prompts per block) against pack + pair: 21.7 / 30.0 / 127 us at
    // R = 16 / 64 / 256 against 19.4 / 19.9 / 35.1 - its per-block reductions grew with R.  This one scales its grid with the bytes
    // (profiles/r06_iou.txt); pack + pair keeps P > 4, resampled or float masks, masks beyond 2 M pixels and R > 4096.
    int G, nb;
    onepass_shape(R, &G, &nb);
    const int GB = G * nb;
    const long long groups = ((long long)R + GB - 1) / GB, chunks = (hw / 32 + OP_CHUNK - 1) / OP_CHUNK;
    if (!g_iou_fused || elem_type != 0 || P > FUSED_MAXP || h != H || w != W || hw % 32 != 0 || chunks > OP_MAX_CHUNKS || groups > OP_TICKETS / 4 ||
        (reinterpret_cast<uintptr_t>(am) & 15) || (reinterpret_cast<uintptr_t>(bm) & 15) ||
        scratch_bytes < mask_iou_fused_scratch_bytes(P, R, hw / 32))
        return false;
    OnepassArgs a;
    a.a = static_cast<const uint8_t*>(am); a.b = static_cast<const uint8_t*>(bm);
    a.hw = hw; a.words = hw / 32; a.P = P; a.R = R;
    a.chunks = (int)chunks; a.groups = (int)groups; a.nb = nb;
    static unsigned next_ticket = 0;  // ring position (per process; every device has its own g_iou_ticket, a range unused there stays zero)
    a.ticket_base = __atomic_fetch_add(&next_ticket, (unsigned)groups, __ATOMIC_RELAXED);
    const bool pk = g_iou_packed && hw < (1ll << 19) && chunks <= 126 && R <= OP_ACC_LINES / 16;
    static unsigned next_acc = 0;
    a.acc_base = pk ? __atomic_fetch_add(&next_acc, 4u * (unsigned)R, __ATOMIC_RELAXED) : 0u;
    a.part = static_cast<unsigned*>(scratch);
    a.inter = inter; a.uni = uni;
    SolaProfScope prof(SOLA_PROF_IOU_PACK, s, 0, (double)(P + R) * hw);
    const dim3 grid((unsigned)chunks, (unsigned)groups);
    if (pk) {
        if (G == 1) hipLaunchKernelGGL((mask_iou_onepass_kernel<1, true>), grid, dim3(256), 0, s, a);
        else if (G == 2) hipLaunchKernelGGL((mask_iou_onepass_kernel<2, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((mask_iou_onepass_kernel<4, true>), grid, dim3(256), 0, s, a);
    } else if (G == 1) hipLaunchKernelGGL(mask_iou_onepass_kernel<1>, grid, dim3(256), 0, s, a);
    else if (G == 2) hipLaunchKernelGGL(mask_iou_onepass_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mask_iou_onepass_kernel<4>, grid, dim3(256), 0, s, a);
    if (hipGetLastError() != hipSuccess) {
        sola_set_error("mask_iou_matrix: one-launch kernel failed to launch");
        *status = SOLA_ERR_HIP;
    }
    return true;
}

// blocks per mask of the resample pack: enough of them to fill the chip when there are few masks, few area atomics when many
static unsigned resample_blocks(int H, int n) {
    const int chunks = (H + RS_RPB - 1) / RS_RPB;
    return (unsigned)std::max(1, std::min(chunks, std::max(4, 2048 / n)));
}

int g_pack_resample_lds = 1;
// Measured (tools/pack_probe.py, -> 540x960): 720x1280 uint8 168 -> 133 us (22 % of the HBM peak on the whole source, of which
// only 3 rows in 4 are read), 1080x1920 uint8 273 -> 151 us (44 %), float32 158 -> 67 us (44 %) and 483 -> 85 us (78 %); rows
// that are not whole aligned 32-pixel groups (480x854) stage pixel by pixel and are slower than the gather kernel: they keep
// it (2 = this kernel for every width: tests).
static bool resample_lds_ok(const void* masks, int elem_type, int h, int w, int W) {
    if (!g_pack_resample_lds || W % 32 != 0 || w > RS_MAXW || w > 3 * W) return false;
    if (g_pack_resample_lds == 2) return true;
    const size_t es = elem_type ? 4 : 1;
    return w % 32 == 0 && (reinterpret_cast<uintptr_t>(masks) & 15) == 0 && ((size_t)h * w * es) % 16 == 0;
}  // sola_tune "pack_resample_lds": 0 = the generic per-pixel kernel for resampled packs (A/B, tests)
void sola_pack_set_resample_lds(int v) { g_pack_resample_lds = v; }

int launch_mask_pack(const void* masks, int elem_type, int n, int h, int w, int H, int W, uint32_t* bits,
                     long long* area, hipStream_t s) {
    SOLA_ARG(n > 0 && h > 0 && w > 0 && H > 0 && W > 0, "mask_pack: bad sizes");
    SOLA_ARG(elem_type == 0 || elem_type == 1, "mask_pack: elem_type %d (0=u8, 1=f32)", elem_type);
    SOLA_ARG(n <= 65535, "mask_pack: n=%d exceeds 65535 masks per call", n);
    PackArgs a;
    a.src = masks; a.bits = bits; a.area = reinterpret_cast<unsigned long long*>(area);
    a.h = h; a.w = w; a.H = H; a.W = W;
    a.hw_src = (long long)h * w; a.HW = (long long)H * W; a.words = (a.HW + 31) / 32;
    a.sy = (float)h / (float)H; a.sx = (float)w / (float)W;
    a.identity = (h == H && w == W) ? 1 : 0;
    a.n_split = n; a.src2 = nullptr; a.bits2 = nullptr; a.area2 = nullptr;
    SOLA_HIP(hipMemsetAsync(area, 0, sizeof(long long) * n, s));
    const long long runs = (a.HW + 15) / 16;
    const unsigned blocks = (unsigned)((runs + 255) / 256);
    const double src_bytes = (double)n * h * w * (elem_type ? 4 : 1);
    SolaProfScope prof(SOLA_PROF_IOU_PACK, s, 0, src_bytes + (double)n * a.words * 4);
    if (elem_type == 0 && a.identity && a.HW % 32 == 0 && (reinterpret_cast<uintptr_t>(masks) & 15) == 0)
        hipLaunchKernelGGL(mask_pack_u8_stream_kernel, dim3((unsigned)((a.words + 1023) / 1024), n), dim3(256), 0, s, a);
    else if (!a.identity && resample_lds_ok(masks, elem_type, h, w, W) && elem_type == 0)
        hipLaunchKernelGGL(mask_pack_resample_kernel<uint8_t>, dim3(resample_blocks(H, n), n), dim3(256), 0, s, a);
    else if (!a.identity && resample_lds_ok(masks, elem_type, h, w, W) && elem_type == 1)
        hipLaunchKernelGGL(mask_pack_resample_kernel<float>, dim3(resample_blocks(H, n), n), dim3(256), 0, s, a);
    else if (elem_type == 0)
        hipLaunchKernelGGL(mask_pack_kernel<uint8_t>, dim3(blocks, n), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(mask_pack_kernel<float>, dim3(blocks, n), dim3(256), 0, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

// Both uint8 mask sets of a P x R matrix at the comparison resolution: ONE memset over the two area arrays (`area_span_bytes` from
// a_area, which must precede b_area in the same buffer) and ONE streaming pack launch instead of two of each - the P = 4, R = 64
// call of the de-dup loop is five dependent stream operations otherwise, and it is their latencies that it takes (bench: iou).
// Returns false when the shapes do not take the streaming kernel (the caller packs the sets separately).
bool launch_mask_pack_pair(const void* a_masks, int P, uint32_t* a_bits, long long* a_area, const void* b_masks, int R, uint32_t* b_bits,
                           long long* b_area, size_t area_span_bytes, int H, int W, hipStream_t s, int* status) {
    const long long HW = (long long)H * W;
    if (HW % 32 != 0 || ((reinterpret_cast<uintptr_t>(a_masks) | reinterpret_cast<uintptr_t>(b_masks)) & 15) != 0 || (HW & 15) != 0 ||
        P + R > 65535 || reinterpret_cast<char*>(b_area) < reinterpret_cast<char*>(a_area))
        return false;
    PackArgs a;
    a.src = a_masks; a.bits = a_bits; a.area = reinterpret_cast<unsigned long long*>(a_area);
    a.src2 = b_masks; a.bits2 = b_bits; a.area2 = reinterpret_cast<unsigned long long*>(b_area); a.n_split = P;
    a.h = H; a.w = W; a.H = H; a.W = W; a.hw_src = HW; a.HW = HW; a.words = (HW + 31) / 32;
    a.sy = 1.f; a.sx = 1.f; a.identity = 1;
    *status = SOLA_OK;
    if (hipMemsetAsync(a_area, 0, area_span_bytes, s) != hipSuccess) {
        sola_set_error("mask_iou_matrix: hipMemsetAsync failed");
        *status = SOLA_ERR_HIP;
        return true;
    }
    SolaProfScope prof(SOLA_PROF_IOU_PACK, s, 0, (double)(P + R) * HW + (double)(P + R) * a.words * 4);
    hipLaunchKernelGGL(mask_pack_u8_stream_kernel, dim3((unsigned)((a.words + 1023) / 1024), P + R), dim3(256), 0, s, a);
    if (hipGetLastError() != hipSuccess) {
        sola_set_error("mask_iou_matrix: pack launch failed");
        *status = SOLA_ERR_HIP;
    }
    return true;
}

int launch_mask_pair(const uint32_t* a_bits, const long long* a_area, int P, int T, const uint32_t* b_bits,
                     const long long* b_area, int R, const int32_t* a_frame, long long words, long long* inter,
                     long long* uni, hipStream_t s) {
    SOLA_ARG(P > 0 && R > 0 && T > 0 && words > 0, "mask_pair: bad sizes");
    PairArgs a{a_bits, b_bits, a_area, b_area, a_frame, words, P, T, R, inter, uni};
    SolaProfScope prof(SOLA_PROF_IOU_PAIR, s, 0, 4.0 * words * ((double)R + (double)R * P));
    hipLaunchKernelGGL(mask_pair_kernel, dim3(R, (P + PB - 1) / PB), dim3(256), 0, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
