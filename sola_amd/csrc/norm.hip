// HBM-bound elementwise / reduction kernels of the track-selection path:
//   ws_standardize  : module/ws.py:9-13   per-out-channel weight standardisation (+ re-layout to [cout][k][cin])
//   group_norm      : nn.GroupNorm at module/module.py:76-92 (encoder, + LeakyReLU) and :34,43,49 (alignment layer)
//   pos_encoding    : module/module.py:112-128
//   lang_concat     : module/module.py:146-147 (text tokens ++ negative tokens) and the mean over W of :152-153
// All reductions are wave-shuffle + LDS trees with a fixed order (deterministic, no atomics).
#include <algorithm>

#include <map>
#include <mutex>
#include <utility>

#include "kernels.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// weight standardisation: one 256-thread block per output channel, the channel's cin*k weights stay in registers
// ---------------------------------------------------------------------------------------------------------------
constexpr int WS_MAX_LAYERS = 8;
constexpr int WS_MAX_PER_THREAD = 16;  // cin*k <= 4096
struct WsArgs {
    WsLayer layer[WS_MAX_LAYERS];
    int first_block[WS_MAX_LAYERS + 1];
    int n_layers;
};

__global__ __launch_bounds__(256) void ws_standardize_kernel(const WsArgs a) {
    __shared__ float red[4];
    int li = 0;
    while (li + 1 < a.n_layers && (int)blockIdx.x >= a.first_block[li + 1]) ++li;
    const WsLayer L = a.layer[li];
    const int co = blockIdx.x - a.first_block[li];
    const int n = L.cin * L.k;
    const float* w = L.w + (long long)co * n;
    float v[WS_MAX_PER_THREAD];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < WS_MAX_PER_THREAD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        v[i] = idx < n ? w[idx] : 0.f;
        s += v[i];
    }
    const float mean = block_sum_256(s, red) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < WS_MAX_PER_THREAD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const float c = idx < n ? v[i] - mean : 0.f;
        v[i] = c;
        q += c * c;
    }
    const float var = block_sum_256(q, red) / (float)(n - 1);  // torch.std: unbiased
    const float denom = sqrtf(var) + 1e-5f;                    // eps is added to the std (ws.py:11)
    float* out = L.out + (long long)co * n;
#pragma unroll
    for (int i = 0; i < WS_MAX_PER_THREAD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < n) {
            const int ci = idx / L.k, kk = idx - ci * L.k;  // source layout [cin][k]
            out[kk * L.cin + ci] = v[i] / denom;            // GEMM layout   [k][cin]
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm over (tokens x channels-of-group): one block per (instance, group); three passes over an L2-resident set
// ---------------------------------------------------------------------------------------------------------------
struct GnArgs {
    const float* x;
    float* y;
    float* y2;
    const float* pe;
    const float *gamma, *beta;
    int inner;
    long long outer_stride, inner_stride, tok_stride;
    int ntok, C, cg, groups;
    float eps, slope;
    int leaky;
    DropoutCfg drop;  // applied after the LeakyReLU (module/module.py:78)
    int out_sp16;     // write y / y2 as split-f16 pairs (cast.hip) instead of f32
    int* guard;       // out_sp16: range guard word (GroupNormDesc::guard), null = unchecked
    const int4* units;  // ragged batches: (first row, row stride, token count, pe row) per instance (GroupNormDesc::units)
    int in_f16, out_f16;  // 16-bit storage mode: x / (y, y2) are _Float16 matrices with the same element offsets
    const float* in_scale_dev;  // optional multiplier applied to x while reading (GroupNormDesc::in_scale_dev)
    void *y_cast, *y2_cast;     // f32 outputs: also their casts for the next GEMM (GroupNormDesc::y_cast)
    int cast_fmt;               // 2 f16, 3 bf16
    float2* stats_out;          // sliced shape only: (mean, rstd) per unit for the backward (GroupNormDesc::stats_out)
};

typedef _Float16 half4n __attribute__((ext_vector_type(4)));
// four consecutive channels starting at element offset `off` of x
__device__ __forceinline__ float4 gn_load(const GnArgs& a, long long off) {
    float4 v;
    if (a.in_f16 == 2) {  // bfloat16 rows (round 6: the bf16 training step's pre-norm activations)
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.x) + off);
        v = make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                        __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
    } else if (a.in_f16) {
        const half4n h = *reinterpret_cast<const half4n*>(reinterpret_cast<const _Float16*>(a.x) + off);
        v = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else {
        v = *reinterpret_cast<const float4*>(a.x + off);
    }
    if (a.in_scale_dev) {
        const float sc = *a.in_scale_dev;
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    }
    return v;
}

// R token slots of a lane -> registers: slot r holds token tl + r * tpp (+ t0) of the unit, zeros beyond ntok.  Written as
// `t < ntok ? gn_load(...) : zero` every slot's load sat in an exec-mask region of its own, behind a format branch, with a full
// s_waitcnt vmcnt(0) before the next one (round 4, code-object scan: R serialized memory round trips per lane).  Here the row index is
// clamped (token 0 of a unit is always there), the loads are unconditional and back to back, the VALUE is selected, and the format
// branch (f32 / f16 input) is taken once around the whole loop.  Same values as gn_load (x * 1.0f is exact where there is no scale).
template <int R>
__device__ __forceinline__ void gn_load_slots(const GnArgs& a, float4 (&v)[R], long long row0, long long tok_stride, int ntok, int t_first, int tpp, int ch) {
    const float sc = a.in_scale_dev ? *a.in_scale_dev : 1.f;
    if (a.in_f16 == 2) {  // bfloat16 rows
        const unsigned short* x = reinterpret_cast<const unsigned short*>(a.x);
        uint2 h[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = t_first + r * tpp;
            h[r] = *reinterpret_cast<const uint2*>(x + (row0 + (long long)(t < ntok ? t : 0) * tok_stride) * a.C + ch);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ok = t_first + r * tpp < ntok;
            v[r] = make_float4(ok ? __builtin_bit_cast(float, h[r].x << 16) * sc : 0.f, ok ? __builtin_bit_cast(float, h[r].x & 0xffff0000u) * sc : 0.f,
                               ok ? __builtin_bit_cast(float, h[r].y << 16) * sc : 0.f, ok ? __builtin_bit_cast(float, h[r].y & 0xffff0000u) * sc : 0.f);
        }
    } else if (a.in_f16) {
        const _Float16* x = reinterpret_cast<const _Float16*>(a.x);
        half4n h[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = t_first + r * tpp;
            h[r] = *reinterpret_cast<const half4n*>(x + (row0 + (long long)(t < ntok ? t : 0) * tok_stride) * a.C + ch);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ok = t_first + r * tpp < ntok;
            v[r] = make_float4(ok ? (float)h[r][0] * sc : 0.f, ok ? (float)h[r][1] * sc : 0.f, ok ? (float)h[r][2] * sc : 0.f, ok ? (float)h[r][3] * sc : 0.f);
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = t_first + r * tpp;
            v[r] = *reinterpret_cast<const float4*>(a.x + (row0 + (long long)(t < ntok ? t : 0) * tok_stride) * a.C + ch);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ok = t_first + r * tpp < ntok;
            v[r] = make_float4(ok ? v[r].x * sc : 0.f, ok ? v[r].y * sc : 0.f, ok ? v[r].z * sc : 0.f, ok ? v[r].w * sc : 0.f);
        }
    }
}

// The token set of instance `inst`: first row, row stride, token count and the positional-encoding row of the y + pe output.
struct GnUnit { long long row0, tok_stride; int ntok, pe_row; };
__device__ __forceinline__ GnUnit gn_unit(const GnArgs& a, int inst) {
    GnUnit u;
    if (a.units) {
        const int4 d = a.units[inst];
        u.row0 = d.x; u.tok_stride = d.y; u.ntok = d.z; u.pe_row = d.w;
    } else {
        u.row0 = (long long)(inst / a.inner) * a.outer_stride + (long long)(inst % a.inner) * a.inner_stride;
        u.tok_stride = a.tok_stride; u.ntok = a.ntok; u.pe_row = inst % a.inner;
    }
    return u;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// normalise + affine (+ LeakyReLU, dropout, the y + pe side output, split-f16 emission) of one float4 and its store
__device__ __forceinline__ void gn_apply_store(const GnArgs& a, long long off, const float4 v, float mean, float rstd, const float4 ga,
                                               const float4 be, const float4 pe, int c4) {
    float4 o;
    o.x = (v.x - mean) * rstd * ga.x + be.x;
    o.y = (v.y - mean) * rstd * ga.y + be.y;
    o.z = (v.z - mean) * rstd * ga.z + be.z;
    o.w = (v.w - mean) * rstd * ga.w + be.w;
    if (a.leaky) {
        o.x = o.x >= 0.f ? o.x : o.x * a.slope;
        o.y = o.y >= 0.f ? o.y : o.y * a.slope;
        o.z = o.z >= 0.f ? o.z : o.z * a.slope;
        o.w = o.w >= 0.f ? o.w : o.w * a.slope;
    }
    if (a.drop.enabled) {
        o.x = dropout_keep(a.drop, (unsigned long long)off) ? o.x * a.drop.scale : 0.f;
        o.y = dropout_keep(a.drop, (unsigned long long)off + 1) ? o.y * a.drop.scale : 0.f;
        o.z = dropout_keep(a.drop, (unsigned long long)off + 2) ? o.z * a.drop.scale : 0.f;
        o.w = dropout_keep(a.drop, (unsigned long long)off + 3) ? o.w * a.drop.scale : 0.f;
    }
    if (a.out_f16) {  // plain f16 output (8-byte stores); the range guard as for the split pairs
        if (off < 0) return;
        half4n h;
        h[0] = (_Float16)o.x; h[1] = (_Float16)o.y; h[2] = (_Float16)o.z; h[3] = (_Float16)o.w;
        *reinterpret_cast<half4n*>(reinterpret_cast<_Float16*>(a.y) + off) = h;
        if (a.y2) {
            h[0] = (_Float16)(o.x + pe.x); h[1] = (_Float16)(o.y + pe.y); h[2] = (_Float16)(o.z + pe.z); h[3] = (_Float16)(o.w + pe.w);
            *reinterpret_cast<half4n*>(reinterpret_cast<_Float16*>(a.y2) + off) = h;
        }
        if (a.guard && !(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))) < 65000.f)) atomicOr(a.guard, 1);
        return;
    }
    if (a.out_sp16) {
        // split-f16 output for the 3 x f16 MFMA GEMMs: lanes c4 and c4^1 hold the two halves of an 8-channel block
        // (every lane of the wave must get here: the shuffles are unconditional)
        const float4 o2 = make_float4(o.x + pe.x, o.y + pe.y, o.z + pe.z, o.w + pe.w);
        const float4 n = make_float4(__shfl_xor(o.x, 1, 64), __shfl_xor(o.y, 1, 64), __shfl_xor(o.z, 1, 64), __shfl_xor(o.w, 1, 64));
        const float4 n2 = make_float4(__shfl_xor(o2.x, 1, 64), __shfl_xor(o2.y, 1, 64), __shfl_xor(o2.z, 1, 64), __shfl_xor(o2.w, 1, 64));
        if ((c4 & 1) == 0 && off >= 0) {
            const float v8[8] = {o.x, o.y, o.z, o.w, n.x, n.y, n.z, n.w};
            half8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) { _Float16 h1, l1; split_f16(v8[j], h1, l1); hi[j] = h1; lo[j] = l1; }
            half8* dst = reinterpret_cast<half8*>(a.y + off);
            dst[0] = hi; dst[1] = lo;
            if (a.guard) {  // |y + pe| <= |y| + 1: one test covers both outputs (NaN fails the comparison too)
                float m = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v8[j]));
                if (!(m < 65000.f)) atomicOr(a.guard, 1);
            }
            if (a.y2) {
                const float w8[8] = {o2.x, o2.y, o2.z, o2.w, n2.x, n2.y, n2.z, n2.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) { _Float16 h1, l1; split_f16(w8[j], h1, l1); hi[j] = h1; lo[j] = l1; }
                half8* dst2 = reinterpret_cast<half8*>(a.y2 + off);
                dst2[0] = hi; dst2[1] = lo;
            }
        }
        return;
    }
    if (a.cast_fmt >= 2 && off >= 0) {  // plain f16 / bf16 operand copies: 8 bytes per lane
        const bool bf = a.cast_fmt == 3;
        auto cv = [&](float x) -> _Float16 { return bf ? __builtin_bit_cast(_Float16, (__bf16)x) : (_Float16)x; };
        half4n h;
        if (a.y_cast) {
            h[0] = cv(o.x); h[1] = cv(o.y); h[2] = cv(o.z); h[3] = cv(o.w);
            *reinterpret_cast<half4n*>(static_cast<_Float16*>(a.y_cast) + off) = h;
        }
        if (a.y2_cast) {
            h[0] = cv(o.x + pe.x); h[1] = cv(o.y + pe.y); h[2] = cv(o.z + pe.z); h[3] = cv(o.w + pe.w);
            *reinterpret_cast<half4n*>(static_cast<_Float16*>(a.y2_cast) + off) = h;
        }
    }
    if (off < 0) return;
    *reinterpret_cast<float4*>(a.y + off) = o;
    if (a.y2) *reinterpret_cast<float4*>(a.y2 + off) = make_float4(o.x + pe.x, o.y + pe.y, o.z + pe.z, o.w + pe.w);
}

// General shape: three passes over an L2-resident unit (any token count).
__global__ __launch_bounds__(256) void group_norm_kernel(const GnArgs a) {
    __shared__ float red[4];
    // block -> (instance, group), the group fastest (blocks in flight together cover all column slices of the same token rows)
    const int inst = (int)(blockIdx.x / a.groups), g = (int)(blockIdx.x % a.groups);
    const int f4 = a.cg >> 2;                 // float4 per token slice
    const int lpt = f4;                       // lanes per token
    const int tpp = 256 / lpt;                // tokens per pass
    const int tl = threadIdx.x / lpt;         // token slot of this thread
    const int c4 = threadIdx.x - tl * lpt;    // float4 index inside the slice
    const GnUnit un = gn_unit(a, inst);
    const long long row0 = un.row0, tok_stride = un.tok_stride;
    const int ntok = un.ntok;
    const int ch = g * a.cg + c4 * 4;
    const bool active = tl < tpp;
    const float cnt = (float)ntok * (float)a.cg;

    float s = 0.f;
    if (active)
        for (int t = tl; t < ntok; t += tpp) {
            const float4 v = gn_load(a, (row0 + (long long)t * tok_stride) * a.C + ch);
            s += (v.x + v.y) + (v.z + v.w);
        }
    const float mean = block_sum_256(s, red) / cnt;
    float q = 0.f;
    if (active)
        for (int t = tl; t < ntok; t += tpp) {
            const float4 v = gn_load(a, (row0 + (long long)t * tok_stride) * a.C + ch);
            const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    const float var = block_sum_256(q, red) / cnt;  // biased, as nn.GroupNorm
    const float rstd = 1.0f / sqrtf(var + a.eps);
    if (!active) return;
    const float4 ga = *reinterpret_cast<const float4*>(a.gamma + ch);
    const float4 be = *reinterpret_cast<const float4*>(a.beta + ch);
    float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.y2) pe = *reinterpret_cast<const float4*>(a.pe + (long long)un.pe_row * a.C + ch);
    for (int t = tl; t < ntok; t += tpp) {
        const long long off = (row0 + (long long)t * tok_stride) * a.C + ch;
        gn_apply_store(a, off, gn_load(a, off), mean, rstd, ga, be, pe, c4);
    }
}

// Sum over a block of NTHR threads (NTHR / 64 waves); every thread gets the result; fixed combination order.
template <int NTHR>
__device__ __forceinline__ float block_sum_n(float v, float* red) {
    if (NTHR == 256) return block_sum_256(v, red);
    v = wave_sum(v);
    __syncthreads();  // protect `red` from a previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NTHR / 64; ++i) t += red[i];
    return t;
}

// Register-resident shape: the whole unit is read ONCE into R float4 per lane, mean and the centred second moment are
// taken from the registers, and the result is written straight out - one HBM read + one write, no re-read.
//   WAVE = true : one wave per unit (units of <= 64*4*R floats: the encoder norms and the motion norm, 1-4 KiB each),
//                 four units per block, shuffle reductions only;
//   WAVE = false: one block per unit (the inter-object norm, 32 KiB, R = 8; the object->language norm, 128 KiB, R = 32).
//                 NTHR = 1024 for the largest units: 16 waves share a 128 KiB unit at 8 float4 per lane instead of 4 waves at 32 -
//                 more loads in flight per unit and twice the waves per CU (173 -> ~100 us per launch at 256 samples: the
//                 256-thread shape moved 515 MB at 3.0 TB/s where the small-unit shapes reach 5.4-5.8).
template <int R, bool WAVE, int NTHR = 256>
__global__ __launch_bounds__(NTHR) void group_norm_reg_kernel(const GnArgs a, long long n_units) {
    __shared__ float red[NTHR / 64];
    const int f4 = a.cg >> 2;
    const int nthr = WAVE ? 64 : NTHR;
    const int tid = WAVE ? (threadIdx.x & 63) : threadIdx.x;
    const long long unit = WAVE ? (long long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long long)blockIdx.x;
    if (WAVE && unit >= n_units) return;  // a whole wave leaves; the WAVE shape has no block-level sync
    const int inst = (int)(unit / a.groups), g = (int)(unit - (long long)inst * a.groups);
    const int tpp = nthr / f4;
    const int tl = tid / f4, c4 = tid - tl * f4;
    const GnUnit un = gn_unit(a, inst);
    const long long row0 = un.row0, tok_stride = un.tok_stride;
    const int ntok = un.ntok;
    const int ch = g * a.cg + c4 * 4;
    const float cnt = (float)ntok * (float)a.cg;
    // the affine parameters (and the positional row) are fetched HERE, beside the unit's rows: behind the two reductions - where they are
    // first used, and where the block barriers pin them - they were one more memory round trip on every block's critical path (round 6)
    const float4 ga = *reinterpret_cast<const float4*>(a.gamma + ch);
    const float4 be = *reinterpret_cast<const float4*>(a.beta + ch);
    float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.y2) pe = *reinterpret_cast<const float4*>(a.pe + (long long)un.pe_row * a.C + ch);
    float4 v[R];
    gn_load_slots<R>(a, v, row0, tok_stride, ntok, tl, tpp, ch);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s += (v[r].x + v[r].y) + (v[r].z + v[r].w);
    const float mean = (WAVE ? wave_sum(s) : block_sum_n<NTHR>(s, red)) / cnt;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (tl + r * tpp < ntok) {
            const float d0 = v[r].x - mean, d1 = v[r].y - mean, d2 = v[r].z - mean, d3 = v[r].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    const float var = (WAVE ? wave_sum(q) : block_sum_n<NTHR>(q, red)) / cnt;  // biased, as nn.GroupNorm
    const float rstd = 1.0f / sqrtf(var + a.eps);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = tl + r * tpp;
        // out-of-range token slots still take part in the split-f16 shuffles; off < 0 marks "do not store"
        const long long off = t < ntok ? (row0 + (long long)t * tok_stride) * a.C + ch : -1;
        gn_apply_store(a, off, v[r], mean, rstd, ga, be, pe, c4);
    }
}

// 16-bit storage mode (x, y, y2 plain f16): the register shape above with EIGHT channels per lane - one 16-byte load and one
// 16-byte store per lane and token slot instead of 8-byte ones.  With four channels per lane a unit keeps half as many bytes in
// flight as in the f32 modes and the f16 norms ran at 77 % of the f32 norms' TIME for half their bytes.
// R = token slots of 8 channels per lane; same statistics (two passes over the registers), affine, LeakyReLU, y + pe side output
// and range guard as gn_apply_store's f16 branch.  Inference only (no dropout).
template <int R, bool WAVE, int NTHR = 256>
__global__ __launch_bounds__(NTHR) void group_norm_reg_h8_kernel(const GnArgs a, long long n_units) {
    __shared__ float red[NTHR / 64];
    const int f8 = a.cg >> 3;
    const int nthr = WAVE ? 64 : NTHR;
    const int tid = WAVE ? (threadIdx.x & 63) : threadIdx.x;
    const long long unit = WAVE ? (long long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long long)blockIdx.x;
    if (WAVE && unit >= n_units) return;
    const int inst = (int)(unit / a.groups), g = (int)(unit - (long long)inst * a.groups);
    const int tpp = nthr / f8;
    const int tl = tid / f8, c8 = tid - tl * f8;
    const GnUnit un = gn_unit(a, inst);
    const long long row0 = un.row0, tok_stride = un.tok_stride;
    const int ntok = un.ntok;
    const int ch = g * a.cg + c8 * 8;
    const float cnt = (float)ntok * (float)a.cg;
    const float sc_in = a.in_scale_dev ? *a.in_scale_dev : 1.f;
    const _Float16* x16 = reinterpret_cast<const _Float16*>(a.x);
    float v[R][8];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = tl + r * tpp;
        const bool ok = t < ntok;  // a clamped row is loaded and the VALUE zeroed (no select of pointers)
        const half8 h = *reinterpret_cast<const half8*>(x16 + (row0 + (long long)(ok ? t : 0) * tok_stride) * a.C + ch);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[r][e] = ok ? (float)h[e] * sc_in : 0.f;
        s += ((v[r][0] + v[r][1]) + (v[r][2] + v[r][3])) + ((v[r][4] + v[r][5]) + (v[r][6] + v[r][7]));
    }
    const float mean = (WAVE ? wave_sum(s) : block_sum_n<NTHR>(s, red)) / cnt;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (tl + r * tpp < ntok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[r][e] - mean;
                q += d * d;
            }
        }
    const float var = (WAVE ? wave_sum(q) : block_sum_n<NTHR>(q, red)) / cnt;  // biased, as nn.GroupNorm
    const float rstd = 1.0f / sqrtf(var + a.eps);
    float ga[8], be[8], pe[8];
    {
        const float4 g0 = *reinterpret_cast<const float4*>(a.gamma + ch), g1 = *reinterpret_cast<const float4*>(a.gamma + ch + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(a.beta + ch), b1 = *reinterpret_cast<const float4*>(a.beta + ch + 4);
        ga[0] = g0.x; ga[1] = g0.y; ga[2] = g0.z; ga[3] = g0.w; ga[4] = g1.x; ga[5] = g1.y; ga[6] = g1.z; ga[7] = g1.w;
        be[0] = b0.x; be[1] = b0.y; be[2] = b0.z; be[3] = b0.w; be[4] = b1.x; be[5] = b1.y; be[6] = b1.z; be[7] = b1.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) pe[e] = 0.f;
        if (a.y2) {
            const float4 p0 = *reinterpret_cast<const float4*>(a.pe + (long long)un.pe_row * a.C + ch);
            const float4 p1 = *reinterpret_cast<const float4*>(a.pe + (long long)un.pe_row * a.C + ch + 4);
            pe[0] = p0.x; pe[1] = p0.y; pe[2] = p0.z; pe[3] = p0.w; pe[4] = p1.x; pe[5] = p1.y; pe[6] = p1.z; pe[7] = p1.w;
        }
    }
    float m = 0.f;
    _Float16* y16 = reinterpret_cast<_Float16*>(a.y);
    _Float16* y216 = reinterpret_cast<_Float16*>(a.y2);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = tl + r * tpp;
        if (t < ntok) {
            const long long off = (row0 + (long long)t * tok_stride) * a.C + ch;
            half8 h, h2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o = (v[r][e] - mean) * rstd * ga[e] + be[e];
                if (a.leaky) o = o >= 0.f ? o : o * a.slope;
                h[e] = (_Float16)o;
                h2[e] = (_Float16)(o + pe[e]);
                m = fmaxf(m, fabsf(o));
            }
            *reinterpret_cast<half8*>(y16 + off) = h;
            if (a.y2) *reinterpret_cast<half8*>(y216 + off) = h2;
        }
    }
    if (a.guard && !(m < 65000.f)) atomicOr(a.guard, 1);
}

// Large units (the object->language norm of a long video: N*T' = 2048 tokens x 128 channels = 1 MiB per unit; ragged batches
// whose largest sample has more than 256 tokens): one block per unit walks the unit three times out of L2 and a launch has only
// instances x groups blocks (256 at T = 128, N = 128, 32 samples: one block per CU, ~0.5 ms per launch where the traffic takes
// 0.11).  Here a unit is cut into slices of 32 float4 per thread (256 tokens at 128 channels per group, a FIXED size, so a
// sample's result does not depend on what else is in the batch) and two launches cover (unit, slice):
//   stats : the slice is read once into registers -> (mean_i, M2_i = sum (x - mean_i)^2) -> a slot in scratch;
//   apply : the slots of the unit are combined in slice order (Chan: M2 = sum M2_i + sum n_i (mean_i - mean)^2), then the
//           slice is read again, normalised and stored (same epilogue as every other shape).
// No inter-block waiting.  Traffic 2 reads + 1 write, from (units x slices) blocks.
constexpr int GNC_R = 32;
struct GncGeo { long long row0, tok_stride, unit; int ntok, pe_row, g, inst, t0, tl, c4, tpp, nslice, sl; };
__device__ __forceinline__ GncGeo gnc_geo(const GnArgs& a, int S) {
    GncGeo q;
    // block -> (instance, slice, group), the GROUP fastest: neighbouring blocks read the eight column slices of the same token rows, as the
    // register shapes do (with the slice fastest, the blocks in flight all sat on one 512-byte column slice of 4-KiB rows: 2.8 TB/s)
    const long long rest = blockIdx.x / a.groups;
    q.g = (int)(blockIdx.x - rest * a.groups);
    q.inst = (int)(rest / S);
    const int sl = (int)(rest - (long long)q.inst * S);
    q.sl = sl;
    q.unit = (long long)q.inst * a.groups + q.g;  // slot index of (unit, slice): unit * S + slice
    const int f4 = a.cg >> 2;
    const GnUnit un = gn_unit(a, q.inst);
    q.row0 = un.row0; q.tok_stride = un.tok_stride; q.ntok = un.ntok; q.pe_row = un.pe_row;
    q.tpp = 256 / f4;
    q.tl = threadIdx.x / f4; q.c4 = threadIdx.x - q.tl * f4;
    const int ts = q.tpp * GNC_R;  // tokens per slice
    q.nslice = (q.ntok + ts - 1) / ts;
    q.t0 = sl < q.nslice ? sl * ts : -1;
    return q;
}
__global__ __launch_bounds__(256) void group_norm_slice_stats_kernel(const GnArgs a, int S, float2* __restrict__ slots) {
    __shared__ float red[4];
    const GncGeo q = gnc_geo(a, S);
    if (q.t0 < 0) return;  // block-uniform: this unit has fewer slices
    const int ch = q.g * a.cg + q.c4 * 4;
    float4 v[GNC_R];
    gn_load_slots<GNC_R>(a, v, q.row0, q.tok_stride, q.ntok, q.t0 + q.tl, q.tpp, ch);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < GNC_R; ++r) s += (v[r].x + v[r].y) + (v[r].z + v[r].w);
    const int ntok_i = min(q.ntok - q.t0, q.tpp * GNC_R);
    const float mean = block_sum_256(s, red) / ((float)ntok_i * (float)a.cg);
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < GNC_R; ++r)
        if (q.t0 + q.tl + r * q.tpp < q.ntok) {
            const float d0 = v[r].x - mean, d1 = v[r].y - mean, d2 = v[r].z - mean, d3 = v[r].w - mean;
            m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    m2 = block_sum_256(m2, red);
    // One-slice units (the short samples of a ragged batch) used to be normalised HERE from the registers: that epilogue's 32 calls of
    // gn_apply_store were not unrolled, v[] became a scratch array (528 bytes per lane) and every load of the loop above was followed by a
    // wait and a scratch store - 206 us for 168 MB (round 4, tools/co_regs.py).  The apply launch takes every unit: same numbers
    // (one slice: weight 1, d = 0), the slice's second read comes out of L2.
    if (threadIdx.x == 0) slots[q.unit * S + q.sl] = make_float2(mean, m2);
}
__global__ __launch_bounds__(256) void group_norm_slice_apply_kernel(const GnArgs a, int S, const float2* __restrict__ slots) {
    const GncGeo q = gnc_geo(a, S);
    if (q.t0 < 0) return;
    const int ts = q.tpp * GNC_R;
    const long long unit = q.unit;
    // combine the unit's slices in index order (every block of the unit computes the same numbers).  Lane i of every wave fetches slot i - one
    // round trip instead of 2 * nslice dependent ones in front of the block's first data load - and the sums run over lane broadcasts in
    // the same order as before (same bits).  Units of more than 64 slices (16 384 tokens) keep the loop over memory.
    float mean = 0.f;
    const float ntot = (float)q.ntok * (float)a.cg;
    float m2 = 0.f;
    if (q.nslice <= 64) {
        const int li = threadIdx.x & 63;
        const float2 mine = li < q.nslice ? slots[unit * S + li] : make_float2(0.f, 0.f);
        for (int i = 0; i < q.nslice; ++i) mean += __shfl(mine.x, i, 64) * ((float)min(q.ntok - i * ts, ts) * (float)a.cg / ntot);
        for (int i = 0; i < q.nslice; ++i) {
            const float d = __shfl(mine.x, i, 64) - mean;
            m2 += __shfl(mine.y, i, 64) + (float)min(q.ntok - i * ts, ts) * (float)a.cg * d * d;
        }
    } else {
        for (int i = 0; i < q.nslice; ++i) mean += slots[unit * S + i].x * ((float)min(q.ntok - i * ts, ts) * (float)a.cg / ntot);
        for (int i = 0; i < q.nslice; ++i) {
            const float2 sm = slots[unit * S + i];
            const float d = sm.x - mean;
            m2 += sm.y + (float)min(q.ntok - i * ts, ts) * (float)a.cg * d * d;
        }
    }
    const float rstd = 1.0f / sqrtf(m2 / ntot + a.eps);  // biased variance, as nn.GroupNorm
    if (a.stats_out && q.sl == 0 && threadIdx.x == 0) a.stats_out[unit] = make_float2(mean, rstd);
    const int ch = q.g * a.cg + q.c4 * 4;
    const float4 ga = *reinterpret_cast<const float4*>(a.gamma + ch);
    const float4 be = *reinterpret_cast<const float4*>(a.beta + ch);
    float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.y2) pe = *reinterpret_cast<const float4*>(a.pe + (long long)q.pe_row * a.C + ch);
    // eight slots at a time: their loads back to back (gn_load_slots), then the eight normalise + store steps (each load used to sit behind
    // the previous slot's store with a full wait in between: 249 us for 672 MB on the 128-sample ragged batch)
    constexpr int CH = 8;
    for (int c = 0; c < GNC_R / CH; ++c) {
        const int tf = q.t0 + q.tl + c * CH * q.tpp;
        if (tf - q.tl >= q.ntok) break;  // block-uniform: the slice ends before this chunk
        float4 v[CH];
        gn_load_slots<CH>(a, v, q.row0, q.tok_stride, q.ntok, tf, q.tpp, ch);
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int t = tf + r * q.tpp;
            // out-of-range token slots still take part in the split-f16 shuffles; off < 0 marks "do not store"
            const long long off = t < q.ntok ? (q.row0 + (long long)t * q.tok_stride) * a.C + ch : -1;
            gn_apply_store(a, off, v[r], mean, rstd, ga, be, pe, q.c4);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void pos_encoding_kernel(const float* __restrict__ gauss, int half, int t_len, float max_len, float* pe) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t_len * half) return;
    const int t = i / half, j = i - t * half;
    // module.py:123-127: (t / max_len) @ G, then * 2*pi (the python double is applied as an fp32 scalar), sin | cos
    const float arg = ((float)t / max_len) * gauss[j] * 6.283185307179586f;
    pe[(long long)t * 2 * half + j] = sinf(arg);
    pe[(long long)t * 2 * half + half + j] = cosf(arg);
}

// lang_cat[b, w, :] and lbar[b, :] = mean_w lang_cat[b, w, :]; one thread per (b, float4 column)
// (eight rows' loads in flight per thread: at one sample per call the 256 threads of the only block walked the 48 rows one load at a time)
__global__ void lang_concat_kernel(const float* __restrict__ lang, const float* __restrict__ neg, float* __restrict__ out,
                                   float* __restrict__ lbar, int B, int L, int n_neg, int D) {
    const int d4 = D >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * d4) return;
    const int b = i / d4, c = (i - b * d4) * 4;
    const int W = L + n_neg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w0 = 0; w0 < W; w0 += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int w = min(w0 + j, W - 1);
            v[j] = w < L ? *reinterpret_cast<const float4*>(lang + ((long long)b * L + w) * D + c)
                         : *reinterpret_cast<const float4*>(neg + (long long)(w - L) * D + c);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (w0 + j < W) {  // the same order of additions as a row-by-row walk
                *reinterpret_cast<float4*>(out + ((long long)b * W + w0 + j) * D + c) = v[j];
                acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
            }
        }
    }
    const float inv = 1.f / (float)W;
    *reinterpret_cast<float4*>(lbar + (long long)b * D + c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// round 5: the same concatenation WITHOUT repeating the negative tokens per sample - out = [B * L text rows | n_neg negative rows]; the key /
// value projections of the negative tokens are the same for every sample (module/module.py:146-147 repeats the table), so the projection
// GEMM takes B * L + n_neg rows instead of B * (L + n_neg) and the attention reads the shared rows (AttnDesc::k_private).  lbar: unchanged
// (the mean over the sample's L + n_neg tokens, summed in the same order).
__global__ void lang_concat_shared_kernel(const float* __restrict__ lang, const float* __restrict__ neg, float* out, float* lbar, int B, int L,
                                          int n_neg, int D) {
    const int d4 = D >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * d4) return;
    const int b = i / d4, c = (i - b * d4) * 4;
    const int W = L + n_neg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < W; ++w) {
        const float4 v = w < L ? *reinterpret_cast<const float4*>(lang + ((long long)b * L + w) * D + c)
                               : *reinterpret_cast<const float4*>(neg + (long long)(w - L) * D + c);
        if (w < L) *reinterpret_cast<float4*>(out + ((long long)b * L + w) * D + c) = v;
        else if (b == 0) *reinterpret_cast<float4*>(out + ((long long)B * L + (w - L)) * D + c) = v;
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float inv = 1.f / (float)W;
    *reinterpret_cast<float4*>(lbar + (long long)b * D + c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// ragged batches: sample b reads L_b text rows from lang[units[b].x ..] and writes W_b = L_b + n_neg rows at out[units[b].z ..]
__global__ void lang_concat_ragged_kernel(const float* __restrict__ lang, const float* __restrict__ neg, float* out, float* lbar,
                                          int B, const int4* __restrict__ units, int n_neg, int D) {
    const int d4 = D >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * d4) return;
    const int b = i / d4, c = (i - b * d4) * 4;
    const int4 u = units[b];
    const int L = u.y, W = L + n_neg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < W; ++w) {
        const float4 v = w < L ? *reinterpret_cast<const float4*>(lang + ((long long)u.x + w) * D + c)
                               : *reinterpret_cast<const float4*>(neg + (long long)(w - L) * D + c);
        *reinterpret_cast<float4*>(out + ((long long)u.z + w) * D + c) = v;
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float inv = 1.f / (float)W;
    *reinterpret_cast<float4*>(lbar + (long long)b * D + c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// dst rows list[i].x + r  <-  src rows list[i].y + r, r < list[i].z: the per-video activations repeated for every expression
// that is scored against the video (HBM copy, 16 bytes per lane)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          const int4* __restrict__ list, int row_f4) {
    const int4 e = list[blockIdx.y];
    const long long n = (long long)e.z * row_f4;
    const float4* s4 = reinterpret_cast<const float4*>(src) + (long long)e.y * row_f4;
    float4* d4 = reinterpret_cast<float4*>(dst) + (long long)e.x * row_f4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) d4[i] = s4[i];
}

}  // namespace

int launch_lang_concat_shared(const float* lang, const float* neg, float* out, float* lbar, int B, int L, int n_neg, int D, hipStream_t s) {
    SOLA_ARG(D % 4 == 0 && B > 0 && L > 0 && n_neg >= 0, "lang_concat_shared: bad arguments");
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 4.0 * ((double)B * (2.0 * L + n_neg) + n_neg) * D);
    const int total = B * (D >> 2);
    hipLaunchKernelGGL(lang_concat_shared_kernel, dim3((total + 255) / 256), dim3(256), 0, s, lang, neg, out, lbar, B, L, n_neg, D);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_lang_concat_ragged(const float* lang, const float* neg, float* out, float* lbar, int B, const int4* units, int n_neg,
                              int D, hipStream_t s) {
    SOLA_ARG(D % 4 == 0 && B > 0 && n_neg >= 0 && units, "lang_concat_ragged: bad arguments");
    const int n = B * (D / 4);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 0);
    hipLaunchKernelGGL(lang_concat_ragged_kernel, dim3((n + 255) / 256), dim3(256), 0, s, lang, neg, out, lbar, B, units, n_neg, D);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_gather_rows(const float* src, float* dst, const int4* list, int n, int row_floats, long long total_rows, hipStream_t s) {
    SOLA_ARG(src && dst && list && n > 0 && row_floats % 4 == 0, "gather_rows: bad arguments");
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 8.0 * total_rows * row_floats);
    const long long per = (total_rows / n + 1) * (row_floats / 4);
    const unsigned bx = (unsigned)std::max<long long>(1, std::min<long long>(64, (per + 1023) / 1024));
    hipLaunchKernelGGL(gather_rows_kernel, dim3(bx, n), dim3(256), 0, s, src, dst, list, row_floats / 4);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_ws_standardize(const WsLayer* layers, int n_layers, hipStream_t s) {
    SOLA_ARG(n_layers >= 1 && n_layers <= WS_MAX_LAYERS, "ws: %d layers", n_layers);
    WsArgs a;
    a.n_layers = n_layers;
    int blocks = 0;
    double elems = 0;
    for (int i = 0; i < n_layers; ++i) {
        SOLA_ARG(layers[i].cin * layers[i].k <= 256 * WS_MAX_PER_THREAD && layers[i].cin * layers[i].k >= 2,
                 "ws: cin*k = %d unsupported (2..%d)", layers[i].cin * layers[i].k, 256 * WS_MAX_PER_THREAD);
        a.layer[i] = layers[i];
        a.first_block[i] = blocks;
        blocks += layers[i].cout;
        elems += (double)layers[i].cout * layers[i].cin * layers[i].k;
    }
    a.first_block[n_layers] = blocks;
    SolaProfScope prof(SOLA_PROF_WS, s, 6.0 * elems, 8.0 * elems);
    hipLaunchKernelGGL(ws_standardize_kernel, dim3(blocks), dim3(256), 0, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int g_gn_variant = 1;  // 0 = always the three-pass kernel (A/B), 1 = register-resident shapes where the unit fits
int g_gn_wide = 1;     // sola_tune "gn_wide": 1024-thread blocks for units of 64-128 KiB (A/B)
void sola_gn_set_wide(int v) { g_gn_wide = v; }
int g_gn_slices = 1;   // sola_tune "gn_slices": 0 = three-pass kernel for units that do not fit the registers (A/B)
void sola_gn_set_slices(int v) { g_gn_slices = v; }
// Library-owned scratch of the sliced shape for callers that pass none (the per-stage entry point; the forward orchestrators hand
// over a piece of the caller's workspace): one buffer per (device, stream), so launches on different streams never share slots - a
// stream's launches are ordered among themselves - and a mutex around the table.  Allocated on first use, grown when a launch
// needs more (after that STREAM has drained: nobody else uses the buffer); never during stream capture (the launch then takes the
// three-pass kernel).
static float2* gn_slice_scratch(size_t bytes, hipStream_t s) {
    struct Slot { float2* buf = nullptr; size_t cap = 0; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Slot> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    Slot& sl = table[std::make_pair(dev, s)];
    if (sl.cap >= bytes) return sl.buf;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    if (sl.buf) { (void)hipStreamSynchronize(s); (void)hipFree(sl.buf); sl.buf = nullptr; sl.cap = 0; }
    const size_t want = std::max<size_t>(bytes * 2, (size_t)1 << 20);
    if (hipMalloc(&sl.buf, want) != hipSuccess) { sl.buf = nullptr; return nullptr; }
    sl.cap = want;
    return sl.buf;
}
int g_gn_h8 = 1;  // sola_tune "gn_h8": 0 = four channels per lane in the 16-bit storage mode too (A/B)
void sola_gn_set_h8(int v) { g_gn_h8 = v; }
void sola_gn_set_variant(int v) { g_gn_variant = v; }

int launch_group_norm(const GroupNormDesc& d, hipStream_t s) {
    SOLA_ARG(d.groups > 0 && d.C % d.groups == 0, "group_norm: C=%d groups=%d", d.C, d.groups);
    const int cg = d.C / d.groups;
    SOLA_ARG(cg % 4 == 0 && cg / 4 <= 256, "group_norm: channels per group %d must be a multiple of 4 and <= 1024", cg);
    SOLA_ARG(d.n_inst > 0 && d.ntok > 0 && d.inner > 0 && d.groups <= 65535, "group_norm: bad sizes");
    SOLA_ARG(d.y2 == nullptr || d.pe != nullptr, "group_norm: y2 needs pe");
    GnArgs a;
    a.x = d.x; a.y = d.y; a.y2 = d.y2; a.pe = d.pe; a.gamma = d.gamma; a.beta = d.beta;
    a.inner = d.inner; a.outer_stride = d.outer_stride; a.inner_stride = d.inner_stride; a.tok_stride = d.tok_stride;
    a.ntok = d.ntok; a.C = d.C; a.cg = cg; a.groups = d.groups; a.eps = d.eps; a.slope = d.slope; a.leaky = d.leaky; a.drop = d.drop; a.out_sp16 = d.out_sp16; a.guard = (d.out_sp16 || d.out_f16) ? d.guard : nullptr; a.units = d.units;
    a.in_f16 = d.in_f16; a.out_f16 = d.out_f16; a.in_scale_dev = d.in_scale_dev;
    a.y_cast = d.y_cast; a.y2_cast = d.y2_cast; a.cast_fmt = (d.y_cast || d.y2_cast) ? d.cast_fmt : 0;
    a.stats_out = static_cast<float2*>(d.stats_out);
    SOLA_ARG(a.cast_fmt == 0 || (!d.out_sp16 && !d.out_f16 && a.cast_fmt >= 2 && a.cast_fmt <= 3 && (!d.y2_cast || d.y2)),
             "group_norm: operand casts go with f32 outputs (format 2 = f16, 3 = bf16)");
    SOLA_ARG(!(d.out_f16 && d.out_sp16), "group_norm: one output format at a time");
    SOLA_ARG(!d.out_sp16 || cg % 8 == 0, "group_norm: split-f16 output needs channels per group %% 8 == 0");
    const double elems = (double)d.n_inst * d.ntok * d.C;
    SolaProfScope prof(SOLA_PROF_NORM, s, 8.0 * elems, (d.y2 ? 12.0 : 8.0) * elems);
    const int f4 = cg / 4;
    const long long n_units = (long long)d.n_inst * d.groups;
    SOLA_ARG(n_units < (1ll << 31), "group_norm: %lld (instance, group) units exceed the grid", n_units);
    const int rw = 64 % f4 == 0 ? (d.ntok + 64 / f4 - 1) / (64 / f4) : 1 << 30;     // float4 per lane, one wave per unit
    const int rb = 256 % f4 == 0 ? (d.ntok + 256 / f4 - 1) / (256 / f4) : 1 << 30;  // ... one block per unit
    // 16-bit storage mode: eight channels (16 bytes) per lane where the unit fits the register shapes
    const int f8 = cg / 8;
    const bool h8 = g_gn_variant != 0 && g_gn_h8 && d.in_f16 && d.out_f16 && !d.drop.enabled && cg % 8 == 0 && n_units < (1ll << 31);
    const int rw8 = (h8 && 64 % f8 == 0) ? (d.ntok + 64 / f8 - 1) / (64 / f8) : 1 << 30;
    const int rb8 = (h8 && 256 % f8 == 0) ? (d.ntok + 256 / f8 - 1) / (256 / f8) : 1 << 30;
    const int rk8 = (h8 && 1024 % f8 == 0) ? (d.ntok + 1024 / f8 - 1) / (1024 / f8) : 1 << 30;
    if (rw8 <= 2) {
        const dim3 grid((unsigned)((n_units + 3) / 4));
        if (rw8 == 1) hipLaunchKernelGGL((group_norm_reg_h8_kernel<1, true>), grid, dim3(256), 0, s, a, n_units);
        else hipLaunchKernelGGL((group_norm_reg_h8_kernel<2, true>), grid, dim3(256), 0, s, a, n_units);
    } else if (rb8 <= 4) {
        const dim3 grid((unsigned)n_units);
        if (rb8 <= 2) hipLaunchKernelGGL((group_norm_reg_h8_kernel<2, false>), grid, dim3(256), 0, s, a, n_units);
        else hipLaunchKernelGGL((group_norm_reg_h8_kernel<4, false>), grid, dim3(256), 0, s, a, n_units);
    } else if (rk8 <= 4) {
        hipLaunchKernelGGL((group_norm_reg_h8_kernel<4, false, 1024>), dim3((unsigned)n_units), dim3(1024), 0, s, a, n_units);
    } else if (g_gn_variant != 0 && rw <= 4 && n_units < (1ll << 31)) {
        const dim3 grid((unsigned)((n_units + 3) / 4));
        if (rw == 1) hipLaunchKernelGGL((group_norm_reg_kernel<1, true>), grid, dim3(256), 0, s, a, n_units);
        else if (rw == 2) hipLaunchKernelGGL((group_norm_reg_kernel<2, true>), grid, dim3(256), 0, s, a, n_units);
        else hipLaunchKernelGGL((group_norm_reg_kernel<4, true>), grid, dim3(256), 0, s, a, n_units);
    } else if (g_gn_variant != 0 && rb <= 32 && n_units < (1ll << 31)) {
        const dim3 grid((unsigned)n_units);
        if (rb <= 2) hipLaunchKernelGGL((group_norm_reg_kernel<2, false>), grid, dim3(256), 0, s, a, n_units);
        else if (rb <= 4) hipLaunchKernelGGL((group_norm_reg_kernel<4, false>), grid, dim3(256), 0, s, a, n_units);
        else if (rb <= 8) hipLaunchKernelGGL((group_norm_reg_kernel<8, false>), grid, dim3(256), 0, s, a, n_units);
        else if (rb <= 16 || !g_gn_wide || 1024 % f4 != 0) {
            if (rb <= 16) hipLaunchKernelGGL((group_norm_reg_kernel<16, false>), grid, dim3(256), 0, s, a, n_units);
            else hipLaunchKernelGGL((group_norm_reg_kernel<32, false>), grid, dim3(256), 0, s, a, n_units);
        } else {
            hipLaunchKernelGGL((group_norm_reg_kernel<8, false, 1024>), grid, dim3(1024), 0, s, a, n_units);  // rb in (16, 32]: 8 per lane at 1024 threads
        }
    } else if (g_gn_variant != 0 && g_gn_slices && 256 % f4 == 0 && n_units < (1ll << 24)) {
        // units of more than 32 float4 per thread: (unit, slice) blocks, two launches (stats, apply)
        const int ts = (256 / f4) * GNC_R;
        const int S = (d.ntok + ts - 1) / ts;
        const long long blocks = n_units * S;
        float2* slots = (d.slice_ws && d.slice_ws_bytes >= (size_t)blocks * sizeof(float2)) ? static_cast<float2*>(d.slice_ws)
                                                                                         : gn_slice_scratch((size_t)blocks * sizeof(float2), s);
        if (slots && blocks < (1ll << 31)) {
            hipLaunchKernelGGL(group_norm_slice_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, S, slots);
            SOLA_LAUNCH_CHECK();
            hipLaunchKernelGGL(group_norm_slice_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, S, slots);
            if (d.stats_out && d.stats_written) *d.stats_written = 1;
        } else {
            hipLaunchKernelGGL(group_norm_kernel, dim3((unsigned)n_units), dim3(256), 0, s, a);
        }
    } else {
        hipLaunchKernelGGL(group_norm_kernel, dim3((unsigned)n_units), dim3(256), 0, s, a);
    }
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_pos_encoding(const float* gauss, int D, int t_len, int max_len, float* pe, hipStream_t s) {
    SOLA_ARG(D % 2 == 0 && t_len > 0, "pos_encoding: D=%d t_len=%d", D, t_len);
    const int half = D / 2, n = t_len * half;
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 4.0 * (n * 2 + half));
    hipLaunchKernelGGL(pos_encoding_kernel, dim3((n + 255) / 256), dim3(256), 0, s, gauss, half, t_len, (float)max_len, pe);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_lang_concat(const float* lang, const float* neg, float* out, float* lbar, int B, int L, int n_neg, int D,
                       hipStream_t s) {
    SOLA_ARG(D % 4 == 0 && B > 0 && L >= 0 && n_neg >= 0 && L + n_neg > 0, "lang_concat: bad sizes");
    const int n = B * (D / 4);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 8.0 * B * (L + n_neg) * D);
    hipLaunchKernelGGL(lang_concat_kernel, dim3((n + 255) / 256), dim3(256), 0, s, lang, neg, out, lbar, B, L, n_neg, D);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
