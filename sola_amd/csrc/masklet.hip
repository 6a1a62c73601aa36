// Masklet resampling and decoding on the bit-packed mask representation of iou.hip.
//
//   mask_bilinear_pack : seg_utils.py:145-160 reshape_masklet — F.interpolate(mode='bilinear', align_corners=False)
//                        followed by `> 0.5` — fused with the 1 bit/pixel pack and the area popcount, so the fp32
//                        [T,540,960] intermediate of the reference never exists.  Source index / weights follow ATen
//                        (UpSample.h area_pixel_compute_source_index + guard_index_and_lambda):
//                            src = max(scale * (dst + 0.5) - 0.5, 0)   (one fused multiply-add, as the CPU build does)
//                            i0 = min(int(src), in-1); l1 = clamp(src - i0, 0, 1); i1 = i0 + (i0 < in-1); l0 = 1 - l1
//                            out = l0y * (l0x*v00 + l1x*v01) + l1y * (l0x*v10 + l1x*v11)
//                        For {0,1} inputs the `> 0.5` decision does not depend on how the sums are fused
//                        (tests/test_masklet_oracle.py checks all 16 corner patterns at every output pixel).
//   mask_unpack        : bits -> {0,1} uint8 / float32 images (the float tensor reshape_masklet returns).
//   rle_fill_or        : COCO run-length masks (column-major runs, pycocotools rleDecode) of K selected tracks OR-ed
//                        into one row-major frame (dataloader.py:305-369 get_sam2_masklet / rle_masklet_decode).
#include <algorithm>

#include "kernels.h"

namespace {

struct BilinearArgs {
    const void* src;
    uint32_t* bits;
    unsigned long long* area;
    int h, w, H, W;
    long long hw_src;
    long long words;
    float sy, sx;
    int rows_per_block;  // a block packs this many destination rows (one area atomic per block)
};

__device__ __forceinline__ void source_index(float scale, int dst, int in, int out, int& i0, float& l1) {
    if (in == out) {  // ATen copies when the size is unchanged
        i0 = dst; l1 = 0.f;
        return;
    }
    float s = __fmaf_rn(scale, (float)dst + 0.5f, -0.5f);
    s = s < 0.f ? 0.f : s;
    i0 = min((int)s, in - 1);
    l1 = fminf(fmaxf(__fsub_rn(s, (float)i0), 0.f), 1.f);
}

// MODE 0: uint8, non-zero -> 1.0;  1: float32 value as is;  2: float32 tracker logits, (v > 0) -> 1.0
// (generate_tokens_grid.py:215-222 `(out_mask_logits > 0.0).float()` folded into the read).
template <typename T, int MODE>
__device__ __forceinline__ float mask_value(T v) {
    if constexpr (MODE == 0) return v ? 1.f : 0.f;
    else if constexpr (MODE == 1) return v;
    else return v > 0.f ? 1.f : 0.f;
}

// A block packs `rows_per_block` destination rows of one mask.  The per-column source index and weight are the same
// for every row, so they are computed once per block into LDS; a row's source rows and weights are wave-uniform.  A wave
// owns 64 consecutive destination pixels of the row per iteration, one pixel per lane, and gathers the decisions by a
// ballot.  Neighbouring lanes read neighbouring (or identical) source elements, so the four taps are coalesced row
// segments.  ALIGNED (W % 32 == 0): the ballot is two whole words of the packed row; otherwise the 64 bits straddle up
// to three words shared with other waves and are OR-ed into a pre-zeroed row atomically.
// STAGED (w % 4 == 0): the source rows the block's destination rows touch are first streamed into LDS with 16-byte
// (4-byte for u8) loads, binarised to one byte per pixel on the way, and the four taps come from LDS - the scattered
// per-pixel dword loads of the direct version were bound by the L1/TA rate (~12 cycles per wave-load), not by HBM.
// Plain float32 masks (MODE 1) are staged too: a block that meets a value other than 0 or 1 while staging falls back
// to the direct float taps for its rows, so arbitrary float images keep ATen's arithmetic.
template <typename T, int MODE, bool ALIGNED, bool STAGED>
__global__ __launch_bounds__(256) void mask_bilinear_pack_kernel(const BilinearArgs a) {
    extern __shared__ int2 xtab[];  // [W] (i0, bits of l1), then the staged rows
    __shared__ int red[4];
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int x = threadIdx.x; x < a.W; x += 256) {
        int i0;
        float l1;
        source_index(a.sx, x, a.w, a.W, i0, l1);
        xtab[x] = make_int2(i0, __float_as_int(l1));
    }
    const T* src = reinterpret_cast<const T*>(a.src) + (long long)n * a.hw_src;
    uint32_t* dst = a.bits + (long long)n * a.words;
    int cnt = 0;
    const int y_begin = blockIdx.x * a.rows_per_block;
    const int y_end = min(a.H, y_begin + a.rows_per_block);
    uint8_t* rows8 = reinterpret_cast<uint8_t*>(xtab + a.W);
    int ys0 = 0;
    int not_binary = 0;
    if (STAGED) {
        int yl;
        float unused;
        source_index(a.sy, y_begin, a.h, a.H, ys0, unused);
        source_index(a.sy, y_end - 1, a.h, a.H, yl, unused);
        const int nrows = yl + (yl < a.h - 1 ? 1 : 0) - ys0 + 1;
        const int wq = a.w >> 2;
        uint32_t* rows32 = reinterpret_cast<uint32_t*>(rows8);
        for (int idx = threadIdx.x; idx < nrows * wq; idx += 256) {
            const int r = idx / wq, c = idx - r * wq;
            uint32_t packed;
            if constexpr (sizeof(T) == 4) {
                const float4 v = reinterpret_cast<const float4*>(src + (long long)(ys0 + r) * a.w)[c];
                if (MODE == 1)
                    not_binary |= (v.x != 0.f && v.x != 1.f) | (v.y != 0.f && v.y != 1.f) | (v.z != 0.f && v.z != 1.f) | (v.w != 0.f && v.w != 1.f);
                packed = (uint32_t)(mask_value<float, MODE>(v.x) != 0.f) | (uint32_t)(mask_value<float, MODE>(v.y) != 0.f) << 8 |
                         (uint32_t)(mask_value<float, MODE>(v.z) != 0.f) << 16 | (uint32_t)(mask_value<float, MODE>(v.w) != 0.f) << 24;
            } else {
                const uint32_t wv = reinterpret_cast<const uint32_t*>(src + (long long)(ys0 + r) * a.w)[c];
                packed = ((((wv & 0x7f7f7f7fu) + 0x7f7f7f7fu) | wv) & 0x80808080u) >> 7;  // byte != 0 -> 1
            }
            rows32[idx] = packed;
        }
    }
    const int any_not_binary = __syncthreads_or(not_binary);  // the barrier also publishes xtab and the staged rows
    const bool from_lds = STAGED && any_not_binary == 0;
    for (int y = y_begin; y < y_end; ++y) {
        int y0;
        float ly1;
        source_index(a.sy, y, a.h, a.H, y0, ly1);
        const float ly0 = __fsub_rn(1.f, ly1);
        const T* r0 = src + (long long)y0 * a.w;
        const T* r1 = r0 + (y0 < a.h - 1 ? a.w : 0);
        const uint8_t* s0 = rows8 + (y0 - ys0) * a.w;
        const uint8_t* s1 = s0 + (y0 < a.h - 1 ? a.w : 0);
        const long long rowbit = (long long)y * a.W;
        for (int x0 = wave * 64; x0 < a.W; x0 += 256) {
            const int x = x0 + lane;
            const bool ok = x < a.W;
            const int2 e = xtab[ok ? x : 0];
            const int i0 = e.x, i1 = e.x + (e.x < a.w - 1 ? 1 : 0);
            const float lx1 = __int_as_float(e.y), lx0 = __fsub_rn(1.f, lx1);
            float v00, v01, v10, v11;
            if (from_lds) {
                v00 = (float)s0[i0]; v01 = (float)s0[i1]; v10 = (float)s1[i0]; v11 = (float)s1[i1];
            } else {
                v00 = mask_value<T, MODE>(r0[i0]); v01 = mask_value<T, MODE>(r0[i1]);
                v10 = mask_value<T, MODE>(r1[i0]); v11 = mask_value<T, MODE>(r1[i1]);
            }
            const float t = __fmaf_rn(lx0, v00, __fmul_rn(lx1, v01));
            const float u = __fmaf_rn(lx0, v10, __fmul_rn(lx1, v11));
            const float o = __fmaf_rn(ly0, t, __fmul_rn(ly1, u));
            const unsigned long long m = __ballot(ok && o > 0.5f);
            cnt += __popcll(m);
            if (lane == 0) {
                const long long p0 = rowbit + x0;
                const long long word = p0 >> 5;
                if (ALIGNED) {
                    dst[word] = (uint32_t)m;
                    if (x0 + 32 < a.W) dst[word + 1] = (uint32_t)(m >> 32);
                } else if (m) {
                    const int sh = (int)(p0 & 31);
                    const uint32_t w0 = (uint32_t)(m << sh), w1 = (uint32_t)(m >> (32 - sh));
                    const uint32_t w2 = sh ? (uint32_t)(m >> (64 - sh)) : 0u;
                    if (w0) atomicOr(dst + word, w0);
                    if (w1) atomicOr(dst + word + 1, w1);
                    if (w2) atomicOr(dst + word + 2, w2);
                }
            }
        }
    }
    if (lane == 0) red[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = red[0] + red[1] + red[2] + red[3];
        if (tot) atomicAdd(&a.area[n], (unsigned long long)tot);
    }
}

template <typename T>
struct Vec4;
template <>
struct Vec4<float> { using type = float4; };
template <>
struct Vec4<uint8_t> { using type = uchar4; };

// One lane expands 4 consecutive pixels (they share a word because 4 | 32).
template <typename T>
__global__ __launch_bounds__(256) void mask_unpack_kernel(const uint32_t* __restrict__ bits, T* __restrict__ out,
                                                          long long HW, long long words, int vec_ok) {
    const int n = blockIdx.y;
    const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= HW) return;
    const uint32_t wv = bits[(long long)n * words + (p0 >> 5)] >> (p0 & 31);
    T* o = out + (long long)n * HW + p0;
    if (vec_ok && p0 + 4 <= HW) {
        typename Vec4<T>::type q;
        q.x = (T)(wv & 1u); q.y = (T)((wv >> 1) & 1u); q.z = (T)((wv >> 2) & 1u); q.w = (T)((wv >> 3) & 1u);
        *reinterpret_cast<typename Vec4<T>::type*>(o) = q;
    } else {
        for (int i = 0; i < 4 && p0 + i < HW; ++i) o[i] = (T)((wv >> i) & 1u);
    }
}

struct RleArgs {
    const uint32_t* cum;    // inclusive prefix sums of the run lengths, all masks back to back
    const long long* off;   // [n_frames*K + 1] first run of mask (frame*K + k); an empty range = an absent frame
    uint8_t* out;           // [n_frames, h, w] row-major {0,1}, may be null
    uint32_t* bits;         // [n_frames, words] packed, may be null
    unsigned long long* area;  // [n_frames], with bits
    int K, h, w;
    int HW;
    long long words;
};

// value of COCO-RLE mask at column-major position pos: runs alternate 0,1,0,... so it is the parity of the number of
// run ends <= pos (upper bound in the prefix sums; zero-length runs are handled by the same rule).
__device__ __forceinline__ unsigned rle_value(const uint32_t* __restrict__ cum, long long lo, long long hi, uint32_t pos) {
    const long long first = lo;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (cum[mid] <= pos) lo = mid + 1; else hi = mid;
    }
    return (unsigned)((lo - first) & 1);
}

__global__ __launch_bounds__(256) void rle_fill_or_kernel(const RleArgs a) {
    __shared__ int red[4];
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;  // row-major pixel
    unsigned val = 0;
    if (p < a.HW) {
        const int y = (int)p / a.w, x = (int)p - y * a.w;
        const uint32_t pos = (uint32_t)x * (uint32_t)a.h + (uint32_t)y;
        for (int k = 0; k < a.K && !val; ++k) {
            const long long m = (long long)f * a.K + k;
            val |= rle_value(a.cum, a.off[m], a.off[m + 1], pos);
        }
        if (a.out) a.out[(long long)f * a.HW + p] = (uint8_t)val;
    }
    if (a.bits) {
        const unsigned long long m = __ballot(val != 0);
        if (lane == 0) {
            const long long word = ((long long)blockIdx.x * 256 + wave * 64) >> 5;
            uint32_t* dst = a.bits + (long long)f * a.words;
            if (word < a.words) dst[word] = (uint32_t)m;
            if (word + 1 < a.words) dst[word + 1] = (uint32_t)(m >> 32);
            red[wave] = __popcll(m);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = red[0] + red[1] + red[2] + red[3];
            if (tot) atomicAdd(&a.area[f], (unsigned long long)tot);
        }
    }
}

}  // namespace

template <typename T, int MODE>
static void launch_bilinear(const BilinearArgs& a, int n, bool aligned, size_t staged_bytes, hipStream_t s) {
    const dim3 grid((a.H + a.rows_per_block - 1) / a.rows_per_block, n), block(256);
    const size_t lds = (size_t)a.W * sizeof(int2) + staged_bytes;
    if (staged_bytes) {
        if (aligned) hipLaunchKernelGGL((mask_bilinear_pack_kernel<T, MODE, true, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((mask_bilinear_pack_kernel<T, MODE, false, true>), grid, block, lds, s, a);
    } else {
        if (aligned) hipLaunchKernelGGL((mask_bilinear_pack_kernel<T, MODE, true, false>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((mask_bilinear_pack_kernel<T, MODE, false, false>), grid, block, lds, s, a);
    }
}

int g_bilinear_staged = 1;  // 0 = always the direct-load version, 1 = stage uint8 sources (measured: 77 -> 58 us per 64 720p
                            // frames; float sources gain nothing at 720p and lose at 1080p, their loop is VALU-bound), 2 = stage all
void sola_bilinear_set_staged(int v) { g_bilinear_staged = v; }

int launch_mask_bilinear_pack(const void* masks, int elem_type, int n, int h, int w, int H, int W, uint32_t* bits,
                              long long* area, hipStream_t s) {
    SOLA_ARG(n > 0 && h > 0 && w > 0 && H > 0 && W > 0, "mask_bilinear_pack: bad sizes");
    SOLA_ARG(elem_type >= 0 && elem_type <= 2, "mask_bilinear_pack: elem_type %d (0=u8, 1=f32, 2=f32 logits)", elem_type);
    SOLA_ARG(n <= 65535, "mask_bilinear_pack: n=%d exceeds 65535 masks per call", n);
    SOLA_ARG(W <= 4096, "mask_bilinear_pack: destination width %d exceeds 4096", W);
    SOLA_ARG((long long)H * W < (1ll << 31) && (long long)h * w < (1ll << 31), "mask_bilinear_pack: image too large");
    BilinearArgs a;
    a.src = masks; a.bits = bits; a.area = reinterpret_cast<unsigned long long*>(area);
    a.h = h; a.w = w; a.H = H; a.W = W;
    a.hw_src = (long long)h * w; a.words = ((long long)H * W + 31) / 32;
    a.sy = (float)h / (float)H; a.sx = (float)w / (float)W;
    // ~16 blocks per CU over the whole launch (measured optimum 2K-8K blocks for 64 frames of 540x960): more blocks only
    // add same-address area atomics, which are the scarce resource
    a.rows_per_block = (int)std::min<long long>(H, std::max<long long>(1, ((long long)H * n + 4095) / 4096));
    // staged version: the source rows of a block (rows_per_block * h/H + 2, one byte per pixel) must fit beside the x table
    size_t staged_bytes = 0;
    const int esz = elem_type ? 4 : 1;
    if ((g_bilinear_staged == 2 || (g_bilinear_staged == 1 && elem_type == 0)) && w % 4 == 0 && (reinterpret_cast<uintptr_t>(masks) & (uintptr_t)(4 * esz - 1)) == 0) {
        a.rows_per_block = std::max(a.rows_per_block, std::min(H, 4));
        for (;;) {
            const long long src_rows = (long long)a.rows_per_block * h / H + 3;
            staged_bytes = (size_t)src_rows * w;
            if ((size_t)W * sizeof(int2) + staged_bytes <= 60 * 1024) break;
            if (a.rows_per_block == 1) { staged_bytes = 0; break; }
            a.rows_per_block = std::max(1, a.rows_per_block / 2);
        }
    }
    SOLA_HIP(hipMemsetAsync(area, 0, sizeof(long long) * n, s));
    const bool aligned = W % 32 == 0;
    if (!aligned) SOLA_HIP(hipMemsetAsync(bits, 0, sizeof(uint32_t) * (size_t)n * a.words, s));
    SolaProfScope prof(SOLA_PROF_IOU_PACK, s, 0, (double)n * h * w * esz + (double)n * a.words * 4);
    if (elem_type == 0) launch_bilinear<uint8_t, 0>(a, n, aligned, staged_bytes, s);
    else if (elem_type == 1) launch_bilinear<float, 1>(a, n, aligned, staged_bytes, s);
    else launch_bilinear<float, 2>(a, n, aligned, staged_bytes, s);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_mask_unpack(const uint32_t* bits, int n, int H, int W, void* out, int elem_type, hipStream_t s) {
    SOLA_ARG(n > 0 && H > 0 && W > 0, "mask_unpack: bad sizes");
    SOLA_ARG(elem_type == 0 || elem_type == 1, "mask_unpack: elem_type %d (0=u8, 1=f32)", elem_type);
    SOLA_ARG(n <= 65535, "mask_unpack: n=%d exceeds 65535 masks per call", n);
    const long long HW = (long long)H * W, words = (HW + 31) / 32;
    const unsigned blocks = (unsigned)((HW + 1023) / 1024);
    const int esz = elem_type ? 4 : 1;
    const int vec_ok = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & (uintptr_t)(4 * esz - 1)) == 0);
    SolaProfScope prof(SOLA_PROF_IOU_PACK, s, 0, (double)n * HW * esz + (double)n * words * 4);
    if (elem_type == 0)
        hipLaunchKernelGGL(mask_unpack_kernel<uint8_t>, dim3(blocks, n), dim3(256), 0, s, bits, reinterpret_cast<uint8_t*>(out), HW, words, vec_ok);
    else
        hipLaunchKernelGGL(mask_unpack_kernel<float>, dim3(blocks, n), dim3(256), 0, s, bits, reinterpret_cast<float*>(out), HW, words, vec_ok);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_rle_fill_or(const uint32_t* cum, const long long* off, int n_frames, int K, int h, int w, uint8_t* out,
                       uint32_t* bits, long long* area, hipStream_t s) {
    SOLA_ARG(n_frames > 0 && K > 0 && h > 0 && w > 0, "rle_fill_or: bad sizes");
    SOLA_ARG(n_frames <= 65535, "rle_fill_or: n_frames=%d exceeds 65535 per call", n_frames);
    SOLA_ARG((long long)h * w < (1ll << 31), "rle_fill_or: image too large");
    SOLA_ARG(out || bits, "rle_fill_or: no output requested");
    SOLA_ARG(!bits || area, "rle_fill_or: packed output needs the area array");
    RleArgs a;
    a.cum = cum; a.off = off; a.out = out; a.bits = bits; a.area = reinterpret_cast<unsigned long long*>(area);
    a.K = K; a.h = h; a.w = w; a.HW = h * w; a.words = ((long long)a.HW + 31) / 32;
    if (bits) SOLA_HIP(hipMemsetAsync(area, 0, sizeof(long long) * n_frames, s));
    SolaProfScope prof(SOLA_PROF_IOU_PACK, s, 0, (double)n_frames * a.HW * (out ? 1 : 0) + (double)n_frames * a.words * (bits ? 4 : 0));
    hipLaunchKernelGGL(rle_fill_or_kernel, dim3((unsigned)(((long long)a.HW + 255) / 256), n_frames), dim3(256), 0, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
