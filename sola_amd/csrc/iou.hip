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
//      16-64 prompts, uint8 masks already at the comparison resolution) ---------------------------------------------------
// The three-kernel path above costs five launches (two area memsets, two packs, one pair kernel): ~25 us of kernels inside
// ~45 us of wall time for 10-35 MB of masks.  Here block (chunk, prompt group) packs its 8 KiB slice of the P track masks
// to bits ONCE (registers), then streams the same slice of up to 16 prompts, eight in flight at a time, and counts
// popc(A & B), |B| (and |A|) without ever writing a bit plane; per-block counts go to a small partial table and the last
// block to finish (one atomic counter) folds them in index order and writes inter / union.  Every mask byte is read once
// from HBM (the tracks again per prompt group, from L2).  One 64-byte memset + one kernel; integer sums, exact.
struct FusedArgs {
    const uint8_t *a, *b;
    long long hw, words;
    int P, R, chunks, groups;
    unsigned* acc;    // [R][5]: inter[p] (p < 4), |B_r| ; then [4]: |A_p|   (zeroed by the caller, with `done`)
    unsigned* done;   // 1 counter
    long long *inter, *uni;
};
constexpr int FUSED_MAXP = 4;
constexpr int FUSED_RG = 16;  // prompts per block
constexpr int FUSED_RB = 8;   // prompts in flight

// sum over the 64 lanes of two 16-bit counts packed in one word (each total < 2^16)
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void mask_iou_fused_u8_kernel(const FusedArgs a) {
    __shared__ unsigned red[4][FUSED_RB][3];
    __shared__ unsigned reda[4][2];
    __shared__ int is_last;
    const int chunk = blockIdx.x, grp = blockIdx.y;
    const long long w = (long long)chunk * 256 + threadIdx.x;
    const bool ok = w < a.words;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned wa[FUSED_MAXP];
#pragma unroll
    for (int p = 0; p < FUSED_MAXP; ++p) {
        wa[p] = 0u;
        if (p < a.P && ok) {
            const uint4* src = reinterpret_cast<const uint4*>(a.a + (long long)p * a.hw);
            wa[p] = pack16(src[2 * w]) | (pack16(src[2 * w + 1]) << 16);
        }
    }
    if (grp == 0) {  // |A_p| of this slice, once
        const unsigned s01 = wave_sum_u32(__popc(wa[0]) | (__popc(wa[1]) << 16));
        const unsigned s23 = wave_sum_u32(__popc(wa[2]) | (__popc(wa[3]) << 16));
        if (lane == 0) { reda[wave][0] = s01; reda[wave][1] = s23; }
    }
    const int r0 = grp * FUSED_RG, r1 = min(a.R, r0 + FUSED_RG);
    for (int rb = r0; rb < r1; rb += FUSED_RB) {
        uint4 lo[FUSED_RB], hi[FUSED_RB];
#pragma unroll
        for (int j = 0; j < FUSED_RB; ++j) {
            const int r = min(rb + j, a.R - 1);
            const uint4* src = reinterpret_cast<const uint4*>(a.b + (long long)r * a.hw);
            lo[j] = ok ? src[2 * w] : make_uint4(0u, 0u, 0u, 0u);
            hi[j] = ok ? src[2 * w + 1] : make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();  // the previous batch's sums have been consumed
#pragma unroll
        for (int j = 0; j < FUSED_RB; ++j) {
            const unsigned wb = pack16(lo[j]) | (pack16(hi[j]) << 16);
            const unsigned s01 = wave_sum_u32(__popc(wa[0] & wb) | (__popc(wa[1] & wb) << 16));
            const unsigned s23 = wave_sum_u32(__popc(wa[2] & wb) | (__popc(wa[3] & wb) << 16));
            const unsigned sb = wave_sum_u32(__popc(wb));
            if (lane == 0) { red[wave][j][0] = s01; red[wave][j][1] = s23; red[wave][j][2] = sb; }
        }
        __syncthreads();
        if (threadIdx.x < FUSED_RB * 5) {
            const int j = threadIdx.x / 5, q = threadIdx.x - j * 5;
            const int r = rb + j;
            if (r < r1 && (q < a.P || q == 4)) {
                unsigned tot = 0;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) {
                    const unsigned v = q == 4 ? red[wv][j][2] : red[wv][j][q >> 1];
                    tot += q == 4 ? v : ((q & 1) ? v >> 16 : v & 0xffffu);
                }
                if (tot) atomicAdd(&a.acc[r * 5 + q], tot);
            }
        }
    }
    if (grp == 0) {
        __syncthreads();
        if (threadIdx.x < a.P) {
            unsigned tot = 0;
            for (int wv = 0; wv < 4; ++wv) {
                const unsigned v = reda[wv][threadIdx.x >> 1];
                tot += (threadIdx.x & 1) ? v >> 16 : v & 0xffffu;
            }
            if (tot) atomicAdd(&a.acc[a.R * 5 + threadIdx.x], tot);
        }
    }
    // ---- the last block to finish writes the matrices (release: the adds above, then the counter; acquire on the other side)
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(a.done, 1u) == (unsigned)(a.chunks * a.groups - 1);
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    for (int i = threadIdx.x; i < a.P * a.R; i += 256) {
        const int p = i / a.R, r = i - p * a.R;
        const long long in = __hip_atomic_load(&a.acc[r * 5 + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long ab = __hip_atomic_load(&a.acc[r * 5 + 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long aa = __hip_atomic_load(&a.acc[a.R * 5 + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.inter[i] = in;
        a.uni[i] = aa + ab - in;  // sum(A + B) - inter (seg_utils.py:133-134)
    }
}

}  // namespace

size_t mask_iou_fused_scratch_bytes(int P, int R, long long words) {
    (void)P; (void)words;
    return 64 + ((size_t)R * 5 + 4) * 4;
}

int g_iou_fused = 1;  // sola_tune "iou_fused": 0 forces the pack + pair path (A/B, tests)
void sola_iou_set_fused(int v) { g_iou_fused = v; }

// true if the call was served by the fused kernel
bool launch_mask_iou_fused(const void* am, const void* bm, int elem_type, int P, int R, int H, int W, int h, int w, long long* inter,
                           long long* uni, void* scratch, size_t scratch_bytes, hipStream_t s, int* status) {
    *status = SOLA_OK;
    const long long hw = (long long)H * W;
    // masks of up to 2^32 - 1 pixels keep every count inside 32 bits; a 256-word slice keeps the packed wave sums inside 16
    // Measured (tools/iou_probe.py, P=4, 540x960): R=16 18.6 us per call fused vs 25.3 us pack + pair; R=64 30.6 vs 25.8; R=256
    // 125 vs 40 (the per-block wave reductions grow with R while the pack + pair path amortises its launches).  Round 3: the pack +
    // pair path packs BOTH mask sets in one launch behind one memset (launch_mask_pack_pair) - three stream operations instead of
    // five - and takes 19.4 / 18.8 / 19.9 / 35.1 us at R = 16 / 32 / 64 / 256 against 21.7 / 21.8 / 30.0 / 127 fused
    // (tools/iou_ab.py): fused up to 16 prompts only (g_iou_fused == 2 forces it for any R: tests).
    if (!g_iou_fused || (R > 16 && g_iou_fused != 2) || elem_type != 0 || P > FUSED_MAXP || h != H || w != W || hw % 32 != 0 ||
        hw > 0xffffffffll || R > 16 * 65535 ||
        (reinterpret_cast<uintptr_t>(am) & 15) || (reinterpret_cast<uintptr_t>(bm) & 15) ||
        scratch_bytes < mask_iou_fused_scratch_bytes(P, R, hw / 32))
        return false;
    FusedArgs a;
    a.a = static_cast<const uint8_t*>(am); a.b = static_cast<const uint8_t*>(bm);
    a.hw = hw; a.words = hw / 32; a.P = P; a.R = R;
    a.chunks = (int)((a.words + 255) / 256);
    a.groups = (R + FUSED_RG - 1) / FUSED_RG;
    a.done = static_cast<unsigned*>(scratch);
    a.acc = a.done + 16;
    a.inter = inter; a.uni = uni;
    if (hipMemsetAsync(scratch, 0, mask_iou_fused_scratch_bytes(P, R, a.words), s) != hipSuccess) {
        sola_set_error("mask_iou_matrix: hipMemsetAsync failed");
        *status = SOLA_ERR_HIP;
        return true;
    }
    SolaProfScope prof(SOLA_PROF_IOU_PACK, s, 0, (double)(P + R) * hw);
    hipLaunchKernelGGL(mask_iou_fused_u8_kernel, dim3(a.chunks, a.groups), dim3(256), 0, s, a);
    if (hipGetLastError() != hipSuccess) {
        sola_set_error("mask_iou_matrix: fused kernel launch failed");
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
