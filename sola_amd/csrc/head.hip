// Score head (module/module.py:152-160), training losses (train.py:98-113 with tools/loss.py:29-56) and the
// inference-time selection (inference.py:59-60).  Everything here is [N, T']-sized after one [T',D].[D] matvec per
// track, so one 256-thread block per (sample, track) with wave-shuffle reductions is enough; the mean over B*N is
// a second single-block kernel with a fixed summation order (deterministic, no atomics).
#include "kernels.h"

namespace {

struct HeadArgs {
    const float* x;
    const float* lbar;
    float* score_map;
    float* score_tokens;
    int N, Tp, D;
    const int4* units;  // ragged batches: (first row, -, T', sample) per track (HeadDesc::units)
};

// score_logits[t] = x[t,:] . mean_w(lang)   (== mean_w(x . lang_w), module.py:152-153)
// a = softmax_t(score_logits); score_tokens = sum_t a_t x[t,:]; score_map = score_tokens . mean_w(lang)
__global__ __launch_bounds__(256) void score_head_kernel(const HeadArgs a) {
    extern __shared__ float sh[];
    float* logits = sh;               // [Tp]
    float* red = sh + ((a.Tp + 3) & ~3);
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < Tp; t += 4) {
        float s = 0.f;
        for (int i = lane; i < d4n; i += 64) {
            const float4 xv = xb[(long long)t * d4n + i], lv = lb[i];
            s += (xv.x * lv.x + xv.y * lv.y) + (xv.z * lv.z + xv.w * lv.w);
        }
        s = wave_sum(s);
        if (lane == 0) logits[t] = s;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = 0; t < Tp; ++t) mx = fmaxf(mx, logits[t]);
    float den = 0.f;
    for (int t = 0; t < Tp; ++t) den += expf(logits[t] - mx);
    __syncthreads();
    for (int t = threadIdx.x; t < Tp; t += 256) logits[t] = expf(logits[t] - mx) / den;
    __syncthreads();
    float part = 0.f;
    float4* out = reinterpret_cast<float4*>(a.score_tokens + (long long)bn * a.D);
    for (int i = threadIdx.x; i < d4n; i += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < Tp; ++t) {
            const float w = logits[t];
            const float4 xv = xb[(long long)t * d4n + i];
            acc.x += xv.x * w; acc.y += xv.y * w; acc.z += xv.z * w; acc.w += xv.w * w;
        }
        out[i] = acc;
        const float4 lv = lb[i];
        part += (acc.x * lv.x + acc.y * lv.y) + (acc.z * lv.z + acc.w * lv.w);
    }
    const float score = block_sum_256(part, red);
    if (threadIdx.x == 0) a.score_map[bn] = score;
}

struct LossArgs {
    const float *score_map, *score_tokens, *labels, *pos, *neg;
    long long neg_batch_stride;
    int N, D, n_neg;
    float pos_w, temp_scale;
    float* terms;
    int32_t* neg_argmax;
    const int32_t* trk_off;  // ragged batches: tracks of sample b = trk_off[b] .. trk_off[b + 1] (LossDesc::trk_off)
};
// first track and track count of sample b
__device__ __forceinline__ void sample_tracks(const LossArgs& a, int b, long long& first, int& count) {
    if (a.trk_off) { first = a.trk_off[b]; count = a.trk_off[b + 1] - a.trk_off[b]; }
    else { first = (long long)b * a.N; count = a.N; }
}

__device__ __forceinline__ float bce_logits(float x, float y) {
    // F.binary_cross_entropy_with_logits: max(x,0) - x*y + log(1 + exp(-|x|))
    return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(256) void loss_terms_kernel(const LossArgs a) {
    extern __shared__ float sh[];
    float* negl = sh;  // [n_neg]
    float* red = sh + ((a.n_neg + 3) & ~3);
    const int bn = blockIdx.x;
    const int b = bn / a.N;
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
    const float pos_logit = block_sum_256(p, red) * a.temp_scale;  // tools/loss.py:29-32
    for (int m = wave; m < a.n_neg; m += 4) {
        const float4* nv = reinterpret_cast<const float4*>(negb + (long long)m * a.D);
        float s = 0.f;
        for (int i = lane; i < d4n; i += 64) {
            const float4 tv = tok[i], v = nv[i];
            s += (tv.x * v.x + tv.y * v.y) + (tv.z * v.z + tv.w * v.w);
        }
        s = wave_sum(s);
        if (lane == 0) negl[m] = s * a.temp_scale;  // tools/loss.py:33-36
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float y = a.labels[bn];
        const float x = a.score_map[bn];
        const float w = y > 0.f ? a.pos_w : 1.f;  // train.py:98-99
        int arg = 0;
        float best = negl[0];
        for (int m = 1; m < a.n_neg; ++m)  // torch.argmax: first maximum (tools/loss.py:40)
            if (negl[m] > best) { best = negl[m]; arg = m; }
        float neg_sum = 0.f;
        for (int m = 0; m < a.n_neg; ++m) neg_sum += bce_logits(negl[m], m == arg ? (1.f - y) : 0.f);
        a.terms[(long long)bn * 3 + 0] = w * bce_logits(x, y);
        a.terms[(long long)bn * 3 + 1] = bce_logits(pos_logit, y);
        a.terms[(long long)bn * 3 + 2] = neg_sum;
        if (a.neg_argmax) a.neg_argmax[bn] = arg;
    }
}

// Same terms, RPB tracks of one sample per block: the n_neg + 1 vectors every track is dotted with (the sample's
// negatives and its positive token) are fetched once per block instead of once per track - the one-track kernel above
// re-read 132 KB of negatives from L2 per 4 KB of tokens and was bound by that.  Token rows wait in LDS.
constexpr int LOSS_RPB = 8;
__global__ __launch_bounds__(256) void loss_terms_rows_kernel(const LossArgs a) {
    extern __shared__ float sh[];
    const int d4n = a.D >> 2;
    float4* tok = reinterpret_cast<float4*>(sh);                      // [RPB][D]
    float* logit = sh + (size_t)LOSS_RPB * a.D;                       // [RPB][n_neg + 1], pos last
    const int b = blockIdx.y, n0 = blockIdx.x * LOSS_RPB;
    long long first;
    int count;
    sample_tracks(a, b, first, count);
    if (n0 >= count) return;  // ragged: the grid is sized by the largest sample
    const int rows = min(LOSS_RPB, count - n0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < LOSS_RPB * d4n; i += 256) {
        const int r = i / d4n, c = i - r * d4n;
        tok[i] = r < rows ? reinterpret_cast<const float4*>(a.score_tokens + (first + n0 + r) * a.D)[c]
                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const float* negb = a.neg + (long long)b * a.neg_batch_stride;
    const int stride = a.n_neg + 1;
    for (int m = wave; m <= a.n_neg; m += 4) {
        const float4* vec = reinterpret_cast<const float4*>(m < a.n_neg ? negb + (long long)m * a.D : a.pos + (long long)b * a.D);
        float s[LOSS_RPB];
#pragma unroll
        for (int r = 0; r < LOSS_RPB; ++r) s[r] = 0.f;
        for (int i = lane; i < d4n; i += 64) {
            const float4 v = vec[i];
#pragma unroll
            for (int r = 0; r < LOSS_RPB; ++r) {
                const float4 tv = tok[r * d4n + i];
                s[r] += (tv.x * v.x + tv.y * v.y) + (tv.z * v.z + tv.w * v.w);
            }
        }
#pragma unroll
        for (int r = 0; r < LOSS_RPB; ++r) {
            const float t = wave_sum(s[r]);
            if (lane == 0) logit[r * stride + m] = t * a.temp_scale;  // tools/loss.py:29-36
        }
    }
    __syncthreads();
    if (threadIdx.x < rows) {
        const int r = threadIdx.x;
        const long long bn = first + n0 + r;
        const float* negl = logit + r * stride;
        const float y = a.labels[bn];
        const float x = a.score_map[bn];
        const float w = y > 0.f ? a.pos_w : 1.f;  // train.py:98-99
        int arg = 0;
        float best = negl[0];
        for (int m = 1; m < a.n_neg; ++m)  // torch.argmax: first maximum (tools/loss.py:40)
            if (negl[m] > best) { best = negl[m]; arg = m; }
        float neg_sum = 0.f;
        for (int m = 0; m < a.n_neg; ++m) neg_sum += bce_logits(negl[m], m == arg ? (1.f - y) : 0.f);
        a.terms[bn * 3 + 0] = w * bce_logits(x, y);
        a.terms[bn * 3 + 1] = bce_logits(negl[a.n_neg], y);
        a.terms[bn * 3 + 2] = neg_sum;
        if (a.neg_argmax) a.neg_argmax[bn] = arg;
    }
}

// The same block shape with the token rows in REGISTERS: the LDS version above reads 8 token float4s from LDS per negative
// float4 it loads (1056 ds_read_b128 per lane, 8.8 GB of LDS traffic at the headline batch: LDS-bound at ~130 us); here
// every wave keeps its lanes' slices of the 8 rows (NI float4 per row) and only the n_neg + 1 vectors stream through.
// Same per-lane summation order, so the logits are bit-identical to the LDS version's.
template <int NI>
__global__ __launch_bounds__(256) void loss_terms_regs_kernel(const LossArgs a) {
    extern __shared__ float sh[];
    float* logit = sh;  // [RPB][n_neg + 1], pos last
    const int d4n = a.D >> 2;
    const int b = blockIdx.y, n0 = blockIdx.x * LOSS_RPB;
    long long first;
    int count;
    sample_tracks(a, b, first, count);
    if (n0 >= count) return;  // ragged: the grid is sized by the largest sample
    const int rows = min(LOSS_RPB, count - n0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 tk[LOSS_RPB][NI];
#pragma unroll
    for (int r = 0; r < LOSS_RPB; ++r)
#pragma unroll
        for (int i = 0; i < NI; ++i)
            tk[r][i] = r < rows ? reinterpret_cast<const float4*>(a.score_tokens + (first + n0 + r) * a.D)[lane + 64 * i]
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* negb = a.neg + (long long)b * a.neg_batch_stride;
    const int stride = a.n_neg + 1;
    // the next vector travels while this one is reduced (at one sample per call a wave's nine iterations each waited for their own load;
    // the loop holds cross-lane operations, so the compiler will not unroll it over a runtime trip count)
    auto vec_of = [&](int m) { return reinterpret_cast<const float4*>(m < a.n_neg ? negb + (long long)m * a.D : a.pos + (long long)b * a.D); };
    float4 vcur[NI], vnext[NI];
    if (wave <= a.n_neg) {
#pragma unroll
        for (int i = 0; i < NI; ++i) vcur[i] = vec_of(wave)[lane + 64 * i];
    }
    for (int m = wave; m <= a.n_neg; m += 4) {
        if (m + 4 <= a.n_neg) {
#pragma unroll
            for (int i = 0; i < NI; ++i) vnext[i] = vec_of(m + 4)[lane + 64 * i];
        }
        float s[LOSS_RPB];
#pragma unroll
        for (int r = 0; r < LOSS_RPB; ++r) s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4 v = vcur[i];
#pragma unroll
            for (int r = 0; r < LOSS_RPB; ++r) {
                const float4 tv = tk[r][i];
                s[r] += (tv.x * v.x + tv.y * v.y) + (tv.z * v.z + tv.w * v.w);
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) vcur[i] = vnext[i];
#pragma unroll
        for (int r = 0; r < LOSS_RPB; ++r) {
            const float t = wave_sum(s[r]);
            if (lane == 0) logit[r * stride + m] = t * a.temp_scale;  // tools/loss.py:29-36
        }
    }
    (void)d4n;
    __syncthreads();
    if (threadIdx.x < rows) {
        const int r = threadIdx.x;
        const long long bn = first + n0 + r;
        const float* negl = logit + r * stride;
        const float y = a.labels[bn];
        const float x = a.score_map[bn];
        const float w = y > 0.f ? a.pos_w : 1.f;  // train.py:98-99
        int arg = 0;
        float best = negl[0];
        for (int m = 1; m < a.n_neg; ++m)  // torch.argmax: first maximum (tools/loss.py:40)
            if (negl[m] > best) { best = negl[m]; arg = m; }
        float neg_sum = 0.f;
        for (int m = 0; m < a.n_neg; ++m) neg_sum += bce_logits(negl[m], m == arg ? (1.f - y) : 0.f);
        a.terms[bn * 3 + 0] = w * bce_logits(x, y);
        a.terms[bn * 3 + 1] = bce_logits(negl[a.n_neg], y);
        a.terms[bn * 3 + 2] = neg_sum;
        if (a.neg_argmax) a.neg_argmax[bn] = arg;
    }
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* terms, int n, int n_neg, float pos_w,
                                                          float align_w, float* loss3) {
    __shared__ float red[4];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        s0 += terms[(long long)i * 3 + 0];
        s1 += terms[(long long)i * 3 + 1];
        s2 += terms[(long long)i * 3 + 2];
    }
    s0 = block_sum_256(s0, red);
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        const float bce = s0 / (float)n;                                               // mean over B*N (train.py:100)
        const float align = pos_w * (s1 / (float)n) + s2 / ((float)n * (float)n_neg);  // tools/loss.py:45-56
        loss3[0] = bce + align_w * align;                                              // train.py:113
        loss3[1] = bce;
        loss3[2] = align;
    }
}

// ragged batches: one block per sample, the means run over the sample's own tracks (the reference's batch size of 1)
__global__ __launch_bounds__(256) void loss_reduce_ragged_kernel(const float* terms, const int32_t* trk_off, int n_neg, float pos_w,
                                                                 float align_w, float* loss3) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int first = trk_off[b], n = trk_off[b + 1] - first;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        s0 += terms[(long long)(first + i) * 3 + 0];
        s1 += terms[(long long)(first + i) * 3 + 1];
        s2 += terms[(long long)(first + i) * 3 + 2];
    }
    s0 = block_sum_256(s0, red);
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        const float bce = s0 / (float)n;
        const float align = pos_w * (s1 / (float)n) + s2 / ((float)n * (float)n_neg);
        loss3[b * 3 + 0] = bce + align_w * align;
        loss3[b * 3 + 1] = bce;
        loss3[b * 3 + 2] = align;
    }
}

__global__ void select_kernel(const float* score, long long n, float thr, float* prob, float* pred) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float p = 1.f / (1.f + expf(-score[i]));
    if (prob) prob[i] = p;
    if (pred) pred[i] = p > thr ? 1.f : 0.f;  // strict > (inference.py:60)
}

}  // namespace

int launch_score_head(const HeadDesc& d, hipStream_t s) {
    SOLA_ARG(d.D % 4 == 0 && d.B > 0 && d.N > 0 && d.Tp > 0, "score_head: bad sizes");
    HeadArgs a{d.x, d.lbar, d.score_map, d.score_tokens, d.N, d.Tp, d.D, d.units};
    const size_t lds = (((size_t)d.Tp + 3) & ~(size_t)3) * 4 + 16;
    SOLA_ARG(lds <= 60000, "score_head: T'=%d too long", d.Tp);
    const double elems = (double)d.B * d.N * d.Tp * d.D;
    SolaProfScope prof(SOLA_PROF_HEAD, s, 4.0 * elems, 4.0 * elems + 4.0 * d.B * d.N * d.D);
    hipLaunchKernelGGL(score_head_kernel, dim3(d.B * d.N), dim3(256), lds, s, a);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}

int launch_loss(const LossDesc& d, hipStream_t s) {
    SOLA_ARG(d.D % 4 == 0 && d.B > 0 && d.N > 0 && d.n_neg > 0 && d.n_neg < 8192, "loss: bad sizes");
    LossArgs a{d.score_map, d.score_tokens, d.labels, d.pos, d.neg, d.neg_batch_stride, d.N, d.D, d.n_neg,
               d.pos_w, d.temp_scale, d.terms, d.neg_argmax, d.trk_off};
    const size_t lds = (((size_t)d.n_neg + 3) & ~(size_t)3) * 4 + 16;
    {
        SolaProfScope prof(SOLA_PROF_HEAD, s, 2.0 * d.B * d.N * (double)d.D * (d.n_neg + 1),
                           4.0 * d.B * d.N * (double)d.D);
        const size_t lds_rows = ((size_t)LOSS_RPB * d.D + (size_t)LOSS_RPB * (d.n_neg + 1)) * 4;
        const size_t lds_logits = (size_t)LOSS_RPB * (d.n_neg + 1) * 4;
        if (d.D == 1024 && lds_logits <= 64 * 1024)
            hipLaunchKernelGGL(loss_terms_regs_kernel<4>, dim3((d.N + LOSS_RPB - 1) / LOSS_RPB, d.B), dim3(256), lds_logits, s, a);
        else if (d.D == 512 && lds_logits <= 64 * 1024)
            hipLaunchKernelGGL(loss_terms_regs_kernel<2>, dim3((d.N + LOSS_RPB - 1) / LOSS_RPB, d.B), dim3(256), lds_logits, s, a);
        else if (lds_rows <= 64 * 1024)
            hipLaunchKernelGGL(loss_terms_rows_kernel, dim3((d.N + LOSS_RPB - 1) / LOSS_RPB, d.B), dim3(256), lds_rows, s, a);
        else {
            SOLA_ARG(!d.trk_off, "loss: ragged batches need D and n_neg that fit the row kernels' LDS");
            hipLaunchKernelGGL(loss_terms_kernel, dim3(d.B * d.N), dim3(256), lds, s, a);
        }
        SOLA_LAUNCH_CHECK();
    }
    {
        SolaProfScope prof(SOLA_PROF_HEAD, s, 0, 12.0 * d.B * d.N);
        if (d.trk_off)
            hipLaunchKernelGGL(loss_reduce_ragged_kernel, dim3(d.B), dim3(256), 0, s, d.terms, d.trk_off, d.n_neg, d.pos_w, d.align_w, d.loss3);
        else
            hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, d.terms, d.B * d.N, d.n_neg, d.pos_w, d.align_w, d.loss3);
        SOLA_LAUNCH_CHECK();
    }
    return SOLA_OK;
}

int launch_select(const float* score, long long n, float thr, float* prob, float* pred, hipStream_t s) {
    SOLA_ARG(n > 0, "select: n=%lld", n);
    SolaProfScope prof(SOLA_PROF_HEAD, s, 0, 12.0 * n);
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, score, n, thr, prob, pred);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
