// Backward of the HBM-bound stages (autograd of module/module.py:76-92,34-49,152-160, module/ws.py:9-13,
// train.py:98-113, tools/loss.py:29-56).  Same decomposition and deterministic reductions as the forward kernels.
#include "kernels.h"
#include <type_traits>

namespace {

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm backward: one block per (instance, group).  Statistics are recomputed exactly as in the forward.
//   g = dy * lrelu'(y) * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat));
//   per-instance partial dgamma[c] = sum_tokens dy' * xhat, dbeta[c] = sum_tokens dy'   (reduced later by colsum)
// ---------------------------------------------------------------------------------------------------------------
struct GnBwdArgs {
    const float *x, *dy, *dy2, *gamma, *beta;
    float *dx, *dgp, *dbp;
    int inner;
    long long outer_stride, inner_stride, tok_stride;
    int ntok, C, cg;
    float eps, slope;
    int leaky;
    DropoutCfg drop;
    const int4* units;  // ragged batches: (first row, row stride, token count, -) per instance (GroupNormBwdDesc::units)
    unsigned short* dx16;  // optional (round 6): dx once more as bfloat16 rows of the same pitch - the operand of the GEMMs that consume it
    int x_bf16;            // round 6: x (the saved pre-norm rows) is a bfloat16 matrix of the same pitch
    const float2* stats;   // round 6, three-pass kernel: the forward's (mean, rstd) per (instance, group) - no statistics walks (GroupNormBwdDesc::stats_in)
    int dy2_bf16;          // round 6: dy2 is a bfloat16 matrix of the same pitch (a branch gradient written by its GEMM as bfloat16 rows)
};
__device__ __forceinline__ float4 gnb_widen(const uint2 w) {
    return make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                       __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
}
__device__ __forceinline__ float4 gnb_load_x(const GnBwdArgs& a, long long off) {
    if (a.x_bf16) {
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.x) + off);
        return make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                           __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(a.x + off);
}
__device__ __forceinline__ void gn_store_dx(const GnBwdArgs& a, long long off, float4 o) {
    *reinterpret_cast<float4*>(a.dx + off) = o;
    if (a.dx16) {
        typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
        bf4 b;
        b[0] = (__bf16)o.x; b[1] = (__bf16)o.y; b[2] = (__bf16)o.z; b[3] = (__bf16)o.w;
        *reinterpret_cast<bf4*>(a.dx16 + off) = b;
    }
}

// token set of instance `inst` (norm.hip: gn_unit)
struct GnBwdUnit { long long row0, tok_stride; int ntok; };
__device__ __forceinline__ GnBwdUnit gn_bwd_unit(const GnBwdArgs& a, int inst) {
    GnBwdUnit u;
    if (a.units) {
        const int4 d = a.units[inst];
        u.row0 = d.x; u.tok_stride = d.y; u.ntok = d.z;
    } else {
        u.row0 = (long long)(inst / a.inner) * a.outer_stride + (long long)(inst % a.inner) * a.inner_stride;
        u.tok_stride = a.tok_stride; u.ntok = a.ntok;
    }
    return u;
}

template <int NTHR>
__device__ __forceinline__ float bwd_block_sum(float v, float* red) {
    if (NTHR == 256) return block_sum_256(v, red);
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NTHR / 64; ++i) t += red[i];
    return t;
}
// NTHR = 1024 (round 3): units beyond the register shapes are the object->language norm's (a sample's N*T' tokens x 128 channels: up
// to 1 MB per (sample, group)); a launch has only samples x groups blocks and its time is the longest block's four walks over
// its unit - four times the threads move the unit four times as fast (515 -> ~200 us per launch on the MeViS-like mix).
template <int NTHR>
__global__ __launch_bounds__(NTHR) void group_norm_bwd_kernel(const GnBwdArgs a) {
    __shared__ float red[NTHR / 64];
    __shared__ float part[NTHR * 8];
    // block -> (instance, group), the GROUP fastest: blocks in flight together read all the column slices of the same token rows (with one
    // group per grid row they all sat on the same 512-byte slice of 4-KiB rows - norm.hip's slice kernels: 2.8 -> 4.5 TB/s from this alone)
    const int n_groups = a.C / a.cg;
    const int inst = (int)(blockIdx.x / n_groups), g = (int)(blockIdx.x % n_groups);
    const int lpt = a.cg >> 2;
    const int tpp = NTHR / lpt;
    const int tl = threadIdx.x / lpt;
    const int c4 = threadIdx.x - tl * lpt;
    const GnBwdUnit un = gn_bwd_unit(a, inst);
    const long long row0 = un.row0, tok_stride = un.tok_stride;
    const int ntok = un.ntok;
    const int ch = g * a.cg + c4 * 4;
    const bool active = tl < tpp;
    const float cnt = (float)ntok * (float)a.cg;

    float mean, rstd;
    if (a.stats) {  // block-uniform: the forward's statistics of this unit (two of the four walks over x are gone)
        const float2 st = a.stats[blockIdx.x];
        mean = st.x; rstd = st.y;
    } else {
        float s = 0.f;
        if (active)
            for (int t = tl; t < ntok; t += tpp) {
                const float4 v = gnb_load_x(a, (row0 + (long long)t * tok_stride) * a.C + ch);
                s += (v.x + v.y) + (v.z + v.w);
            }
        mean = bwd_block_sum<NTHR>(s, red) / cnt;
        float q = 0.f;
        if (active)
            for (int t = tl; t < ntok; t += tpp) {
                const float4 v = gnb_load_x(a, (row0 + (long long)t * tok_stride) * a.C + ch);
                const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        const float var = bwd_block_sum<NTHR>(q, red) / cnt;
        rstd = 1.0f / sqrtf(var + a.eps);
    }

    float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), be = ga;
    if (active) {
        ga = *reinterpret_cast<const float4*>(a.gamma + ch);
        be = *reinterpret_cast<const float4*>(a.beta + ch);
    }
    // dy' of one element quad from the loaded gradient: dropout mask and LeakyReLU slope as the forward applied them
    auto grad_post = [&](long long off, const float4& xh, float4 d) {
        if (a.drop.enabled) {  // y = dropout(lrelu(gn(x))): the mask is regenerated from the element index
            d.x = dropout_keep(a.drop, (unsigned long long)off) ? d.x * a.drop.scale : 0.f;
            d.y = dropout_keep(a.drop, (unsigned long long)off + 1) ? d.y * a.drop.scale : 0.f;
            d.z = dropout_keep(a.drop, (unsigned long long)off + 2) ? d.z * a.drop.scale : 0.f;
            d.w = dropout_keep(a.drop, (unsigned long long)off + 3) ? d.w * a.drop.scale : 0.f;
        }
        if (a.leaky) {
            if (xh.x * ga.x + be.x < 0.f) d.x *= a.slope;
            if (xh.y * ga.y + be.y < 0.f) d.y *= a.slope;
            if (xh.z * ga.z + be.z < 0.f) d.z *= a.slope;
            if (xh.w * ga.w + be.w < 0.f) d.w *= a.slope;
        }
        return d;
    };
    // One walk over the unit's (x, dy) pairs, FOUR token slots per trip: their eight loads go out back to back (slots past the unit re-read
    // the thread's first token and are skipped), the format of x is decided once around the walk.  One slot per trip was a dependent
    // memory round trip per token: a 600-token unit took twenty of them per walk (round 6).
    auto walk = [&](auto xb, auto&& body) {
        constexpr bool XB = decltype(xb)::value;
        constexpr int U = 4;
        for (int t0 = tl; t0 < ntok; t0 += U * tpp) {
            float4 xr[U], dr[U];
            long long offs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * tpp;
                offs[u] = (row0 + (long long)(t < ntok ? t : tl) * tok_stride) * a.C + ch;
                if constexpr (XB) {
                    const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.x) + offs[u]);
                    xr[u] = make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                                        __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
                } else {
                    xr[u] = *reinterpret_cast<const float4*>(a.x + offs[u]);
                }
                dr[u] = *reinterpret_cast<const float4*>(a.dy + offs[u]);
            }
            if (a.dy2 && a.dy2_bf16) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4 e = gnb_widen(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.dy2) + offs[u]));
                    dr[u].x += e.x; dr[u].y += e.y; dr[u].z += e.z; dr[u].w += e.w;
                }
            } else if (a.dy2) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4 e = *reinterpret_cast<const float4*>(a.dy2 + offs[u]);
                    dr[u].x += e.x; dr[u].y += e.y; dr[u].z += e.z; dr[u].w += e.w;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (t0 + u * tpp < ntok) body(offs[u], xr[u], dr[u]);
        }
    };
    float s1 = 0.f, s2 = 0.f;
    float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam;
    auto sums = [&](long long off, const float4& v, const float4& dl) {
        const float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
        const float4 d = grad_post(off, xh, dl);
        dgam.x += d.x * xh.x; dgam.y += d.y * xh.y; dgam.z += d.z * xh.z; dgam.w += d.w * xh.w;
        dbet.x += d.x; dbet.y += d.y; dbet.z += d.z; dbet.w += d.w;
        const float g0 = d.x * ga.x, g1 = d.y * ga.y, g2 = d.z * ga.z, g3 = d.w * ga.w;
        s1 += (g0 + g1) + (g2 + g3);
        s2 += (g0 * xh.x + g1 * xh.y) + (g2 * xh.z + g3 * xh.w);
    };
    if (active) {
        if (a.x_bf16) walk(std::true_type{}, sums);
        else walk(std::false_type{}, sums);
    }
    const float m1 = bwd_block_sum<NTHR>(s1, red) / cnt;
    const float m2 = bwd_block_sum<NTHR>(s2, red) / cnt;
    // per-channel partials: reduce the token slots that share a channel quad
    float* pp = &part[threadIdx.x * 8];
    pp[0] = dgam.x; pp[1] = dgam.y; pp[2] = dgam.z; pp[3] = dgam.w;
    pp[4] = dbet.x; pp[5] = dbet.y; pp[6] = dbet.z; pp[7] = dbet.w;
    __syncthreads();
    if (threadIdx.x < lpt) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < tpp; ++sl) {
            const float* src = &part[(sl * lpt + threadIdx.x) * 8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += src[j];
        }
        const long long o = (long long)inst * a.C + g * a.cg + threadIdx.x * 4;
        *reinterpret_cast<float4*>(a.dgp + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(a.dbp + o) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    if (!active) return;
    auto apply = [&](long long off, const float4& v, const float4& dl) {
        const float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
        const float4 d = grad_post(off, xh, dl);
        float4 o;
        o.x = rstd * (d.x * ga.x - m1 - xh.x * m2);
        o.y = rstd * (d.y * ga.y - m1 - xh.y * m2);
        o.z = rstd * (d.z * ga.z - m1 - xh.z * m2);
        o.w = rstd * (d.w * ga.w - m1 - xh.w * m2);
        gn_store_dx(a, off, o);
    };
    if (a.x_bf16) walk(std::true_type{}, apply);
    else walk(std::false_type{}, apply);
}

// Register-resident shape (as norm.hip's group_norm_reg_kernel): x and dy of the whole unit are read ONCE into 2 R float4 per
// lane, the statistics, the two gradient means and the per-channel partials all come from the registers, dx is written
// straight out - 8 + 4 bytes per element of traffic instead of four reads of x and two of dy through L2 (the three-pass
// kernel above: 78-133 us per launch at 64 samples where the forward's register kernels take 19-55).
//   WAVE = true : one wave per unit (encoder norms and the motion norm, units of <= 4 KiB), four units per block, shuffles only;
//   WAVE = false: one block per unit (the inter-object norm: 32 KiB, R = 8).
//   NTHR = 1024 (block shape only): the 128 KiB object->language units at 8 float4 per lane and tensor (norm.hip's wide shape).
template <int R, bool WAVE, int NTHR = 256>
__global__ __launch_bounds__(NTHR, (NTHR == 256 && !WAVE) ? (R <= 2 ? 8 : (R == 4 ? 6 : 1)) : 1) void group_norm_bwd_reg_kernel(const GnBwdArgs a, long long n_units, int groups) {
    __shared__ float red[NTHR / 64];
    __shared__ float part[WAVE ? 1 : NTHR * 8];
    const int f4 = a.cg >> 2;
    const int nthr = WAVE ? 64 : NTHR;
    const int tid = WAVE ? (threadIdx.x & 63) : threadIdx.x;
    const long long unit = WAVE ? (long long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long long)blockIdx.x;
    if (WAVE && unit >= n_units) return;  // a whole wave leaves; the WAVE shape has no block-level sync
    const int inst = (int)(unit / groups), g = (int)(unit - (long long)inst * groups);
    const int tpp = nthr / f4;
    const int tl = tid / f4, c4 = tid - tl * f4;
    const GnBwdUnit un = gn_bwd_unit(a, inst);
    const long long row0 = un.row0, tok_stride = un.tok_stride;
    const int ntok = un.ntok;
    const int ch = g * a.cg + c4 * 4;
    const float cnt = (float)ntok * (float)a.cg;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    // (R <= 4: the affine parameters beside the unit's rows, not behind the two reductions - one memory round trip less on the block's
    // critical path; the 8-slot shapes have no register to carry them across the reductions)
    float4 ga = z, be = z;
    if constexpr (R <= 4) {
        ga = *reinterpret_cast<const float4*>(a.gamma + ch);
        be = *reinterpret_cast<const float4*>(a.beta + ch);
    }
    float4 xv[R], dv[R];
    float s = 0.f;
    // Token slots beyond the unit load token 0 (always there) and are zeroed by a select: written as `t < ntok ? *(float4*)p : z` each
    // load became FOUR predicated 4-byte loads in an exec-mask region of its own, and the dy2 term a load + wait per slot (round 4,
    // tools/co_loads.py: 64 dword loads at R = 8; the inter-object norm's backward ran at half the speed its traffic allows).
    // The format of x is decided ONCE around the whole batch of loads: with the test inside gnb_load_x every slot's load sat in a basic block
    // of its own and the bfloat16 side waited for each 8-byte load before issuing the next (round 6, tools/co_dis.py: four dependent round
    // trips at R = 4 - the bf16 steps' norms ran slower on 2-byte rows than on 4-byte ones).
    auto load_all = [&](auto xb) {
        constexpr bool XB = decltype(xb)::value;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = tl + r * tpp;
            const bool ok = t < ntok;
            const long long off = (row0 + (long long)(ok ? t : 0) * tok_stride) * a.C + ch;
            float4 xr;
            if constexpr (XB) {
                const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.x) + off);
                xr = make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                                 __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
            } else {
                xr = *reinterpret_cast<const float4*>(a.x + off);
            }
            const float4 dr = *reinterpret_cast<const float4*>(a.dy + off);
            xv[r] = make_float4(ok ? xr.x : 0.f, ok ? xr.y : 0.f, ok ? xr.z : 0.f, ok ? xr.w : 0.f);
            dv[r] = make_float4(ok ? dr.x : 0.f, ok ? dr.y : 0.f, ok ? dr.z : 0.f, ok ? dr.w : 0.f);
        }
    };
    if (a.x_bf16) load_all(std::true_type{});
    else load_all(std::false_type{});
#pragma unroll
    for (int r = 0; r < R; ++r) s += (xv[r].x + xv[r].y) + (xv[r].z + xv[r].w);
    if (NTHR != 1024 && a.dy2 && a.dy2_bf16) {  // (the format decided around the batch of loads, as for x; the 1024-thread shape - the
                                                  // object->language norm's - never has a second gradient and no register for the code)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = tl + r * tpp;
            const bool ok = t < ntok;
            const long long off = (row0 + (long long)(ok ? t : 0) * tok_stride) * a.C + ch;
            const float4 e = gnb_widen(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.dy2) + off));
            dv[r].x += ok ? e.x : 0.f; dv[r].y += ok ? e.y : 0.f; dv[r].z += ok ? e.z : 0.f; dv[r].w += ok ? e.w : 0.f;
        }
    } else if (a.dy2) {  // block-uniform: the second gradient of the inter-object norm (x_obj feeds x_obj + pe too)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = tl + r * tpp;
            const bool ok = t < ntok;
            const long long off = (row0 + (long long)(ok ? t : 0) * tok_stride) * a.C + ch;
            const float4 e = *reinterpret_cast<const float4*>(a.dy2 + off);
            dv[r].x += ok ? e.x : 0.f; dv[r].y += ok ? e.y : 0.f; dv[r].z += ok ? e.z : 0.f; dv[r].w += ok ? e.w : 0.f;
        }
    }
    const float mean = (WAVE ? wave_sum(s) : bwd_block_sum<NTHR>(s, red)) / cnt;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (tl + r * tpp < ntok) {
            const float d0 = xv[r].x - mean, d1 = xv[r].y - mean, d2 = xv[r].z - mean, d3 = xv[r].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    const float var = (WAVE ? wave_sum(q) : bwd_block_sum<NTHR>(q, red)) / cnt;
    const float rstd = 1.0f / sqrtf(var + a.eps);
    if constexpr (R > 4) {
        ga = *reinterpret_cast<const float4*>(a.gamma + ch);
        be = *reinterpret_cast<const float4*>(a.beta + ch);
    }
    float s1 = 0.f, s2 = 0.f;
    float4 dgam = z, dbet = z;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = tl + r * tpp;
        if (t < ntok) {
            const long long off = (row0 + (long long)t * tok_stride) * a.C + ch;
            const float4 xh = make_float4((xv[r].x - mean) * rstd, (xv[r].y - mean) * rstd, (xv[r].z - mean) * rstd, (xv[r].w - mean) * rstd);
            float4 d = dv[r];
            if (a.drop.enabled) {  // y = dropout(lrelu(gn(x))): the mask is regenerated from the element index
                d.x = dropout_keep(a.drop, (unsigned long long)off) ? d.x * a.drop.scale : 0.f;
                d.y = dropout_keep(a.drop, (unsigned long long)off + 1) ? d.y * a.drop.scale : 0.f;
                d.z = dropout_keep(a.drop, (unsigned long long)off + 2) ? d.z * a.drop.scale : 0.f;
                d.w = dropout_keep(a.drop, (unsigned long long)off + 3) ? d.w * a.drop.scale : 0.f;
            }
            if (a.leaky) {
                if (xh.x * ga.x + be.x < 0.f) d.x *= a.slope;
                if (xh.y * ga.y + be.y < 0.f) d.y *= a.slope;
                if (xh.z * ga.z + be.z < 0.f) d.z *= a.slope;
                if (xh.w * ga.w + be.w < 0.f) d.w *= a.slope;
            }
            dgam.x += d.x * xh.x; dgam.y += d.y * xh.y; dgam.z += d.z * xh.z; dgam.w += d.w * xh.w;
            dbet.x += d.x; dbet.y += d.y; dbet.z += d.z; dbet.w += d.w;
            const float g0 = d.x * ga.x, g1 = d.y * ga.y, g2 = d.z * ga.z, g3 = d.w * ga.w;
            s1 += (g0 + g1) + (g2 + g3);
            s2 += (g0 * xh.x + g1 * xh.y) + (g2 * xh.z + g3 * xh.w);
            xv[r] = xh;                            // kept for the dx pass
            dv[r] = make_float4(g0, g1, g2, g3);   // dy' * gamma
        }
    }
    const float m1 = (WAVE ? wave_sum(s1) : bwd_block_sum<NTHR>(s1, red)) / cnt;
    const float m2 = (WAVE ? wave_sum(s2) : bwd_block_sum<NTHR>(s2, red)) / cnt;
    // per-channel partials: add up the token lanes that share a channel quad (fixed order)
    const long long po = (long long)inst * a.C + g * a.cg + c4 * 4;
    if (WAVE) {
        float acc[8] = {dgam.x, dgam.y, dgam.z, dgam.w, dbet.x, dbet.y, dbet.z, dbet.w};
        for (int o = f4; o < 64; o <<= 1)  // lanes tl differ in the bits above log2(f4); f4 is a power of two here
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
        if (tl == 0) {
            *reinterpret_cast<float4*>(a.dgp + po) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(a.dbp + po) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    } else {
        float* pp = &part[threadIdx.x * 8];
        pp[0] = dgam.x; pp[1] = dgam.y; pp[2] = dgam.z; pp[3] = dgam.w;
        pp[4] = dbet.x; pp[5] = dbet.y; pp[6] = dbet.z; pp[7] = dbet.w;
        __syncthreads();
        if (threadIdx.x < f4) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int sl = 0; sl < tpp; ++sl) {
                const float* src = &part[(sl * f4 + threadIdx.x) * 8];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += src[j];
            }
            *reinterpret_cast<float4*>(a.dgp + po) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(a.dbp + po) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = tl + r * tpp;
        if (t < ntok) {
            const long long off = (row0 + (long long)t * tok_stride) * a.C + ch;
            gn_store_dx(a, off, make_float4(rstd * (dv[r].x - m1 - xv[r].x * m2), rstd * (dv[r].y - m1 - xv[r].y * m2),
                                            rstd * (dv[r].z - m1 - xv[r].z * m2), rstd * (dv[r].w - m1 - xv[r].w * m2)));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight-standardisation backward (module/ws.py:9-13): w_hat = c / (sigma + eps), c = w - mean(w),
// sigma = sqrt(sum c^2 / (n-1)).  dc = g / s - (sum g w_hat) c / (s (n-1) sigma);  dw = dc - mean(dc)
// ---------------------------------------------------------------------------------------------------------------
constexpr int WSB_MAX_LAYERS = 8, WSB_MAX_PER_THREAD = 16;
struct WsBwdArgs {
    WsBwdLayer layer[WSB_MAX_LAYERS];
    int first_block[WSB_MAX_LAYERS + 1];
    int n_layers;
};

__global__ __launch_bounds__(256) void ws_backward_kernel(const WsBwdArgs a) {
    __shared__ float red[4];
    int li = 0;
    while (li + 1 < a.n_layers && (int)blockIdx.x >= a.first_block[li + 1]) ++li;
    const WsBwdLayer L = a.layer[li];
    const int co = blockIdx.x - a.first_block[li];
    const int n = L.cin * L.k;
    const float* w = L.w + (long long)co * n;
    const float* gsrc = L.dwstd + (long long)co * n;
    float c[WSB_MAX_PER_THREAD], g[WSB_MAX_PER_THREAD];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < WSB_MAX_PER_THREAD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        c[i] = idx < n ? w[idx] : 0.f;
        if (idx < n) {
            const int ci = idx / L.k, kk = idx - ci * L.k;
            g[i] = gsrc[kk * L.cin + ci];
        } else {
            g[i] = 0.f;
        }
        s += c[i];
    }
    const float mean = block_sum_256(s, red) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < WSB_MAX_PER_THREAD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        c[i] = idx < n ? c[i] - mean : 0.f;
        q += c[i] * c[i];
    }
    const float var = block_sum_256(q, red) / (float)(n - 1);
    const float sigma = sqrtf(var);
    const float sden = sigma + 1e-5f;
    float gw = 0.f;
#pragma unroll
    for (int i = 0; i < WSB_MAX_PER_THREAD; ++i) gw += g[i] * (c[i] / sden);
    const float A = block_sum_256(gw, red);
    const float k2 = sigma > 0.f ? A / (sden * (float)(n - 1) * sigma) : 0.f;
    float ds = 0.f;
#pragma unroll
    for (int i = 0; i < WSB_MAX_PER_THREAD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        g[i] = idx < n ? g[i] / sden - k2 * c[i] : 0.f;  // dc
        ds += g[i];
    }
    const float dmean = block_sum_256(ds, red) / (float)n;
    float* out = L.dw + (long long)co * n;
#pragma unroll
    for (int i = 0; i < WSB_MAX_PER_THREAD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < n) out[idx] = g[i] - dmean;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// score head backward (module/module.py:152-160)
// ---------------------------------------------------------------------------------------------------------------
struct HeadBwdArgs {
    const float *x, *lbar, *d_score, *d_tok;
    float *dx, *dlbar_part;
    int N, Tp, D;
    const int4* units;  // ragged batches: (first row, -, T', sample) per track (HeadBwdDesc::units); Tp = the largest T'
};

__global__ __launch_bounds__(256) void score_head_bwd_kernel(const HeadBwdArgs a) {
    extern __shared__ float sh[];
    const int tp4 = (a.Tp + 3) & ~3;
    float* aw = sh;             // softmax weights a_t
    float* da = sh + tp4;       // da_t, then dlogit_t
    float* red = sh + 2 * tp4;  // 4
    const int bn = blockIdx.x;
    int b = bn / a.N, Tp = a.Tp;
    long long row0 = (long long)bn * a.Tp;
    if (a.units) {
        const int4 u = a.units[bn];
        row0 = u.x; Tp = u.z; b = u.w;
    }
    const int d4n = a.D >> 2;
    const float4* xb = reinterpret_cast<const float4*>(a.x + row0 * a.D);
    const float4* lb = reinterpret_cast<const float4*>(a.lbar + (long long)b * a.D);
    const float4* dt = reinterpret_cast<const float4*>(a.d_tok + (long long)bn * a.D);
    const float ds = a.d_score[bn];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // logits_t = x_t . lbar ; da_t = dtok . x_t with dtok = d_tok + ds * lbar
    for (int t = wave; t < Tp; t += 4) {
        float s = 0.f, u = 0.f;
        for (int i = lane; i < d4n; i += 64) {
            const float4 xv = xb[(long long)t * d4n + i], lv = lb[i], gv = dt[i];
            s += (xv.x * lv.x + xv.y * lv.y) + (xv.z * lv.z + xv.w * lv.w);
            u += (xv.x * (gv.x + ds * lv.x) + xv.y * (gv.y + ds * lv.y)) + (xv.z * (gv.z + ds * lv.z) + xv.w * (gv.w + ds * lv.w));
        }
        s = wave_sum(s);
        u = wave_sum(u);
        if (lane == 0) { aw[t] = s; da[t] = u; }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = 0; t < Tp; ++t) mx = fmaxf(mx, aw[t]);
    float den = 0.f;
    for (int t = 0; t < Tp; ++t) den += expf(aw[t] - mx);
    float sada = 0.f;
    for (int t = 0; t < Tp; ++t) sada += (expf(aw[t] - mx) / den) * da[t];
    __syncthreads();
    for (int t = threadIdx.x; t < Tp; t += 256) {
        const float w = expf(aw[t] - mx) / den;
        aw[t] = w;
        da[t] = w * (da[t] - sada);  // dlogit_t
    }
    __syncthreads();
    float4* dxo = reinterpret_cast<float4*>(a.dx + row0 * a.D);
    float4* dlb = reinterpret_cast<float4*>(a.dlbar_part + (long long)bn * a.D);
    for (int i = threadIdx.x; i < d4n; i += 256) {
        const float4 lv = lb[i], gv = dt[i];
        const float4 dtok = make_float4(gv.x + ds * lv.x, gv.y + ds * lv.y, gv.z + ds * lv.z, gv.w + ds * lv.w);
        float4 tok = make_float4(0.f, 0.f, 0.f, 0.f), acc = tok;
        for (int t = 0; t < Tp; ++t) {
            const float w = aw[t], dl = da[t];
            const float4 xv = xb[(long long)t * d4n + i];
            tok.x += w * xv.x; tok.y += w * xv.y; tok.z += w * xv.z; tok.w += w * xv.w;
            acc.x += dl * xv.x; acc.y += dl * xv.y; acc.z += dl * xv.z; acc.w += dl * xv.w;
            dxo[(long long)t * d4n + i] = make_float4(w * dtok.x + dl * lv.x, w * dtok.y + dl * lv.y,
                                                      w * dtok.z + dl * lv.z, w * dtok.w + dl * lv.w);
        }
        dlb[i] = make_float4(ds * tok.x + acc.x, ds * tok.y + acc.y, ds * tok.z + acc.z, ds * tok.w + acc.w);
    }
    (void)red;
}

// ---------------------------------------------------------------------------------------------------------------
// loss backward (train.py:98-113, tools/loss.py:29-56)
// ---------------------------------------------------------------------------------------------------------------
struct LossBwdArgs {
    const float *score_map, *score_tokens, *labels, *pos, *neg;
    long long neg_batch_stride;
    int B, N, D, n_neg;
    float pos_w, temp_scale, align_w;
    const float* g3;
    float *d_score, *d_tok, *coef;
    const int32_t* trk_off;  // ragged batches (LossBwdDesc::trk_off): tracks of sample b = trk_off[b] .. trk_off[b + 1], g3 is [B][3]
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void loss_bwd_kernel(const LossBwdArgs a) {
    extern __shared__ float sh[];
    float* negl = sh;  // [n_neg] logits, then coefficients
    float* red = sh + ((a.n_neg + 3) & ~3);
    float* sc = red + 4;  // [2]: dpos coefficient
    const int bn = blockIdx.x;
    int b = bn / a.N, count = a.B * a.N;  // the means run over all B*N tracks, or - ragged - over the sample's own
    const float* g3 = a.g3;
    if (a.trk_off) {
        int lo = 0, hi = a.B - 1;  // largest b with trk_off[b] <= bn
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.trk_off[mid] <= bn) lo = mid;
            else hi = mid - 1;
        }
        b = lo;
        count = a.trk_off[b + 1] - a.trk_off[b];
        g3 = a.g3 + 3 * b;
    }
    const int d4n = a.D >> 2;
    const float4* tok = reinterpret_cast<const float4*>(a.score_tokens + (long long)bn * a.D);
    const float4* pos = reinterpret_cast<const float4*>(a.pos + (long long)b * a.D);
    const float* negb = a.neg + (long long)b * a.neg_batch_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float p = 0.f;
    for (int i = threadIdx.x; i < d4n; i += 256) {
        const float4 tv = tok[i], pv = pos[i];
        p += (tv.x * pv.x + tv.y * pv.y) + (tv.z * pv.z + tv.w * pv.w);
    }
    const float pos_logit = block_sum_256(p, red) * a.temp_scale;
    for (int m = wave; m < a.n_neg; m += 4) {
        const float4* nv = reinterpret_cast<const float4*>(negb + (long long)m * a.D);
        float s = 0.f;
#pragma unroll 4
        for (int i = lane; i < d4n; i += 64) {
            const float4 tv = tok[i], v = nv[i];
            s += (tv.x * v.x + tv.y * v.y) + (tv.z * v.z + tv.w * v.w);
        }
        s = wave_sum(s);
        if (lane == 0) negl[m] = s * a.temp_scale;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float bn_count = (float)count;
        const float y = a.labels[bn], x = a.score_map[bn];
        const float w = y > 0.f ? a.pos_w : 1.f;
        const float cb = g3[0] + g3[1];               // d total / d bce
        const float ca = g3[0] * a.align_w + g3[2];   // d total / d align
        a.d_score[bn] = cb * w * (sigmoidf_(x) - y) / bn_count;
        int arg = 0;
        float best = negl[0];
        for (int m = 1; m < a.n_neg; ++m)
            if (negl[m] > best) { best = negl[m]; arg = m; }
        sc[0] = a.temp_scale * ca * a.pos_w * (sigmoidf_(pos_logit) - y) / bn_count;
        for (int m = 0; m < a.n_neg; ++m) {
            const float target = m == arg ? (1.f - y) : 0.f;
            const float cf = a.temp_scale * ca * (sigmoidf_(negl[m]) - target) / (bn_count * (float)a.n_neg);
            negl[m] = cf;
            a.coef[(long long)bn * a.n_neg + m] = cf;
        }
    }
    __syncthreads();
    const float dpos = sc[0];
    float4* out = reinterpret_cast<float4*>(a.d_tok + (long long)bn * a.D);
    for (int i = threadIdx.x; i < d4n; i += 256) {
        const float4 pv = pos[i];
        float4 acc = make_float4(dpos * pv.x, dpos * pv.y, dpos * pv.z, dpos * pv.w);
#pragma unroll 8
        for (int m = 0; m < a.n_neg; ++m) {
            const float cf = negl[m];
            const float4 v = reinterpret_cast<const float4*>(negb + (long long)m * a.D)[i];
            acc.x += cf * v.x; acc.y += cf * v.y; acc.z += cf * v.z; acc.w += cf * v.w;
        }
        out[i] = acc;
    }
}

// d_neg[b][m][:] = sum_n coef[b,n,m] * tok[b,n,:]
__global__ __launch_bounds__(256) void loss_dneg_kernel(const float* __restrict__ coef, const float* __restrict__ tok,
                                                        float* __restrict__ d_neg, int N, int D, int n_neg,
                                                        const int32_t* __restrict__ trk_off) {
    const int b = blockIdx.x / n_neg, m = blockIdx.x % n_neg;
    const int d4n = D >> 2;
    const long long first = trk_off ? trk_off[b] : (long long)b * N;
    if (trk_off) N = trk_off[b + 1] - trk_off[b];
    for (int i = threadIdx.x; i < d4n; i += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int n = 0; n < N; ++n) {
            const float cf = coef[(first + n) * n_neg + m];
            const float4 v = reinterpret_cast<const float4*>(tok + (first + n) * D)[i];
            acc.x += cf * v.x; acc.y += cf * v.y; acc.z += cf * v.z; acc.w += cf * v.w;
        }
        reinterpret_cast<float4*>(d_neg + ((long long)b * n_neg + m) * D)[i] = acc;
    }
}

// d_negw[m][:] = sum_b ( d_lang[b, L+m, :] + dlbar[b, :] / W + d_neg_align[b, m, :] )
__global__ __launch_bounds__(256) void neg_token_grad_kernel(const float* __restrict__ d_lang, const float* __restrict__ dlbar,
                                                             const float* __restrict__ d_neg_align, float* __restrict__ out,
                                                             int B, int L, int n_neg, int D, const int4* __restrict__ units) {
    const int m = blockIdx.x;
    int W = L + n_neg;
    float invw = 1.f / (float)W;
    for (int i = threadIdx.x; i < (D >> 2); i += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < B; ++b) {
            long long neg_row = (long long)b * W + L + m;
            if (units) {  // ragged: sample b has units[b] = (-, L_b, first row of its text ++ negatives, W_b)
                const int4 u = units[b];
                neg_row = (long long)u.z + u.y + m;
                W = u.w;
                invw = 1.f / (float)W;
            }
            if (d_lang) {
                const float4 v = reinterpret_cast<const float4*>(d_lang + neg_row * D)[i];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            if (dlbar) {
                const float4 v = reinterpret_cast<const float4*>(dlbar + (long long)b * D)[i];
                acc.x += v.x * invw; acc.y += v.y * invw; acc.z += v.z * invw; acc.w += v.w * invw;
            }
            if (d_neg_align) {
                const float4 v = reinterpret_cast<const float4*>(d_neg_align + ((long long)b * n_neg + m) * D)[i];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        reinterpret_cast<float4*>(out + (long long)m * D)[i] = acc;
    }
}

// out[seg][c] = sum of in[r][c] over r in [off[seg], off[seg + 1]): the per-sample sums of the per-track partials of a ragged
// batch (fixed order)
__global__ __launch_bounds__(256) void segsum_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          const int32_t* __restrict__ off, int cols) {
    const int seg = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = off[seg], r1 = off[seg + 1];
    float s0 = 0.f, s1 = 0.f;
    int r = r0;
    for (; r + 1 < r1; r += 2) { s0 += in[(long long)r * cols + c]; s1 += in[(long long)(r + 1) * cols + c]; }
    if (r < r1) s0 += in[(long long)r * cols + c];
    out[(long long)seg * cols + c] = s0 + s1;
}

}  // namespace

int launch_segsum_rows(const float* in, float* out, const int32_t* off, int segments, int cols, hipStream_t s) {
    SOLA_ARG(in && out && off && segments > 0 && segments <= 65535 && cols > 0, "segsum_rows: bad arguments");
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 0);
    hipLaunchKernelGGL(segsum_rows_kernel, dim3((cols + 255) / 256, segments), dim3(256), 0, s, in, out, off, cols);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int g_gn_bwd_reg = 1;  // sola_tune "gn_bwd_reg": 0 = three-pass kernel for every shape (A/B)
void sola_gn_set_bwd_reg(int v) { g_gn_bwd_reg = v; }

// which kernel launch_group_norm_bwd picks for units of at most `ntok` tokens: the 1024-thread register shape (17..32 float4 per lane of a
// 256-thread block) has no bfloat16-dy2 code
static bool gn_bwd_takes_reg1024(int ntok, int cg) {
    const int f4 = cg / 4;
    const bool pow2 = (f4 & (f4 - 1)) == 0;
    if (!g_gn_bwd_reg || !pow2 || f4 > 256) return false;
    const int rw = f4 <= 64 ? (ntok + 64 / f4 - 1) / (64 / f4) : 1 << 30;
    const int rb = (ntok + 256 / f4 - 1) / (256 / f4);
    if (rw <= 4 || rb <= 8) return false;
    if (rb <= 16 && 512 % f4 == 0) return false;
    return rb <= 32 && 1024 % f4 == 0;
}
bool group_norm_bwd_dy2_bf16_supported(int ntok, int C, int groups) {
    return groups > 0 && C % groups == 0 && (C / groups) % 4 == 0 && !gn_bwd_takes_reg1024(ntok, C / groups);
}

int launch_group_norm_bwd(const GroupNormBwdDesc& d, hipStream_t s) {
    SOLA_ARG(d.groups > 0 && d.C % d.groups == 0, "group_norm_bwd: C=%d groups=%d", d.C, d.groups);
    const int cg = d.C / d.groups;
    SOLA_ARG(cg % 4 == 0 && cg / 4 <= 256, "group_norm_bwd: channels per group %d unsupported", cg);
    SOLA_ARG(d.n_inst > 0 && d.ntok > 0 && d.inner > 0, "group_norm_bwd: bad sizes");
    GnBwdArgs a;
    a.x = d.x; a.dy = d.dy; a.dy2 = d.dy2; a.gamma = d.gamma; a.beta = d.beta; a.dx = d.dx; a.dgp = d.dgamma_part; a.dbp = d.dbeta_part;
    a.inner = d.inner; a.outer_stride = d.outer_stride; a.inner_stride = d.inner_stride; a.tok_stride = d.tok_stride;
    a.ntok = d.ntok; a.C = d.C; a.cg = cg; a.eps = d.eps; a.slope = d.slope; a.leaky = d.leaky; a.drop = d.drop; a.units = d.units;
    a.dx16 = static_cast<unsigned short*>(d.dx16);
    a.x_bf16 = d.x_bf16;
    a.stats = static_cast<const float2*>(d.stats_in);
    a.dy2_bf16 = (d.dy2 && d.dy2_bf16) ? 1 : 0;
    SOLA_ARG(!a.dy2_bf16 || group_norm_bwd_dy2_bf16_supported(d.ntok, d.C, d.groups), "group_norm_bwd: a bfloat16 dy2 is not read by this shape's kernel");
    const double elems = (double)d.n_inst * d.ntok * d.C;
    SolaProfScope prof(SOLA_PROF_NORM, s, 20.0 * elems, (d.dy2 ? 16.0 : 12.0) * elems);
    const int f4 = cg / 4;
    const bool pow2 = (f4 & (f4 - 1)) == 0;
    const long long n_units = (long long)d.n_inst * d.groups;
    SOLA_ARG(n_units < (1ll << 31), "group_norm_bwd: %lld (instance, group) units exceed the grid", n_units);
    const int rw = pow2 && f4 <= 64 ? (d.ntok + 64 / f4 - 1) / (64 / f4) : 1 << 30;    // float4 per lane and tensor, one wave per unit
    const int rb = pow2 && f4 <= 256 ? (d.ntok + 256 / f4 - 1) / (256 / f4) : 1 << 30;  // ... one block per unit
    if (g_gn_bwd_reg && rw <= 4 && n_units < (1ll << 31)) {
        const dim3 grid((unsigned)((n_units + 3) / 4));
        if (rw == 1) hipLaunchKernelGGL((group_norm_bwd_reg_kernel<1, true>), grid, dim3(256), 0, s, a, n_units, d.groups);
        else if (rw == 2) hipLaunchKernelGGL((group_norm_bwd_reg_kernel<2, true>), grid, dim3(256), 0, s, a, n_units, d.groups);
        else hipLaunchKernelGGL((group_norm_bwd_reg_kernel<4, true>), grid, dim3(256), 0, s, a, n_units, d.groups);
    } else if (g_gn_bwd_reg && rb <= 8 && n_units < (1ll << 31)) {
        const dim3 grid((unsigned)n_units);
        if (rb <= 2) hipLaunchKernelGGL((group_norm_bwd_reg_kernel<2, false>), grid, dim3(256), 0, s, a, n_units, d.groups);
        else if (rb <= 4) hipLaunchKernelGGL((group_norm_bwd_reg_kernel<4, false>), grid, dim3(256), 0, s, a, n_units, d.groups);
        else hipLaunchKernelGGL((group_norm_bwd_reg_kernel<8, false>), grid, dim3(256), 0, s, a, n_units, d.groups);
    } else if (g_gn_bwd_reg && rb <= 16 && pow2 && f4 <= 256 && 512 % f4 == 0 && n_units < (1ll << 31)) {
        // units of up to twice the 256-thread shape (ragged batches: the inter-object norm over up to 128 tracks): 512 threads at 8
        // float4 per lane and tensor - the 1024-thread shape leaves most of its lanes idle on them (407 -> ~220 us per launch)
        hipLaunchKernelGGL((group_norm_bwd_reg_kernel<8, false, 512>), dim3((unsigned)n_units), dim3(512), 0, s, a, n_units, d.groups);
    } else if (g_gn_bwd_reg && rb <= 32 && pow2 && f4 <= 256 && 1024 % f4 == 0 && n_units < (1ll << 31)) {
        hipLaunchKernelGGL((group_norm_bwd_reg_kernel<8, false, 1024>), dim3((unsigned)n_units), dim3(1024), 0, s, a, n_units, d.groups);
    } else if (g_gn_bwd_reg && (1024 % f4) == 0 && f4 <= 256) {
        hipLaunchKernelGGL(group_norm_bwd_kernel<1024>, dim3((unsigned)n_units), dim3(1024), 0, s, a);
    } else {
        hipLaunchKernelGGL(group_norm_bwd_kernel<256>, dim3((unsigned)n_units), dim3(256), 0, s, a);
    }
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_ws_backward(const WsBwdLayer* layers, int n_layers, hipStream_t s) {
    SOLA_ARG(n_layers >= 1 && n_layers <= WSB_MAX_LAYERS, "ws_backward: %d layers", n_layers);
    WsBwdArgs a;
    a.n_layers = n_layers;
    int blocks = 0;
    double elems = 0;
    for (int i = 0; i < n_layers; ++i) {
        SOLA_ARG(layers[i].cin * layers[i].k <= 256 * WSB_MAX_PER_THREAD && layers[i].cin * layers[i].k >= 2, "ws_backward: cin*k unsupported");
        a.layer[i] = layers[i];
        a.first_block[i] = blocks;
        blocks += layers[i].cout;
        elems += (double)layers[i].cout * layers[i].cin * layers[i].k;
    }
    a.first_block[n_layers] = blocks;
    SolaProfScope prof(SOLA_PROF_WS, s, 12.0 * elems, 12.0 * elems);
    hipLaunchKernelGGL(ws_backward_kernel, dim3(blocks), dim3(256), 0, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_score_head_bwd(const HeadBwdDesc& d, hipStream_t s) {
    SOLA_ARG(d.D % 4 == 0 && d.B > 0 && d.N > 0 && d.Tp > 0, "score_head_bwd: bad sizes");
    HeadBwdArgs a{d.x, d.lbar, d.d_score, d.d_tok, d.dx, d.dlbar_part, d.N, d.Tp, d.D, d.units};
    const size_t lds = (2 * (((size_t)d.Tp + 3) & ~(size_t)3) + 4) * sizeof(float);
    SOLA_ARG(lds <= 60000, "score_head_bwd: T'=%d too long", d.Tp);
    const double elems = (double)d.B * d.N * d.Tp * d.D;
    SolaProfScope prof(SOLA_PROF_HEAD, s, 10.0 * elems, 12.0 * elems);
    hipLaunchKernelGGL(score_head_bwd_kernel, dim3(d.B * d.N), dim3(256), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_loss_bwd(const LossBwdDesc& d, hipStream_t s) {
    SOLA_ARG(d.D % 4 == 0 && d.B > 0 && d.N > 0 && d.n_neg > 0 && d.n_neg < 8192, "loss_bwd: bad sizes");
    LossBwdArgs a{d.score_map, d.score_tokens, d.labels, d.pos, d.neg, d.neg_batch_stride, d.B, d.N, d.D, d.n_neg,
                  d.pos_w, d.temp_scale, d.align_w, d.g3, d.d_score, d.d_tok, d.coef, d.trk_off};
    const size_t lds = ((((size_t)d.n_neg + 3) & ~(size_t)3) + 8) * sizeof(float);
    SolaProfScope prof(SOLA_PROF_HEAD, s, 4.0 * d.B * d.N * (double)d.D * (d.n_neg + 1), 8.0 * d.B * d.N * (double)d.D);
    // ragged: d.N = the largest track count (unused by the kernels), the grid covers the concatenated tracks
    const unsigned n_trk = d.trk_off ? (unsigned)d.total_tracks : (unsigned)(d.B * d.N);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(n_trk), dim3(256), lds, s, a);
    SOLA_LAUNCH_CHECK();
    if (d.d_neg) {
        hipLaunchKernelGGL(loss_dneg_kernel, dim3(d.B * d.n_neg), dim3(256), 0, s, d.coef, d.score_tokens, d.d_neg, d.N, d.D, d.n_neg, d.trk_off);
        SOLA_LAUNCH_CHECK();
    }
    return SOLA_OK;
}

int launch_neg_token_grad(const float* d_lang, const float* dlbar, const float* d_neg_align, float* d_negw, int B, int L,
                          int n_neg, int D, hipStream_t s, const int4* units) {
    SOLA_ARG(d_negw && B > 0 && n_neg > 0 && D % 4 == 0, "neg_token_grad: bad arguments");
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, 12.0 * B * n_neg * (double)D);
    hipLaunchKernelGGL(neg_token_grad_kernel, dim3(n_neg), dim3(256), 0, s, d_lang, dlbar, d_neg_align, d_negw, B, L, n_neg, D, units);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
