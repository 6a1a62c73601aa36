// Multi-tensor gradient statistics for the training step:
//   get_grad_norm_dict (module/module.py:164-199): per-group L2 norms of the parameter gradients.  The reference pays
//     one .item() host sync per parameter (83 per step); here one launch reduces every gradient tensor and a second
//     one folds the per-block partials per group in a fixed order (deterministic), so the host reads 8 doubles once.
//   clip_grad_norm_ (train.py:121-122): one launch scales every gradient tensor by a device-side factor.
// HBM-bound streaming over the 131.9 MB of gradients (float4 loads, double accumulation of the block partials).
#include "kernels.h"

namespace {

constexpr int MT_MAX = 128;      // tensors per launch (the path has 83 parameters)
constexpr int MT_CHUNK = 4096;   // floats per block

struct MtArgs {
    const float* ptr[MT_MAX];
    long long numel[MT_MAX];
    int first_block[MT_MAX + 1];
    int group[MT_MAX];
    int n;
};

__device__ __forceinline__ int find_tensor(const MtArgs& a, int block) {
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {  // last tensor whose first_block <= block
        const int mid = (lo + hi + 1) >> 1;
        if (a.first_block[mid] <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void mt_sqnorm_kernel(const MtArgs a, double* __restrict__ partial, int* __restrict__ pgroup) {
    __shared__ float red[4];
    const int t = find_tensor(a, blockIdx.x);
    const long long base = (long long)(blockIdx.x - a.first_block[t]) * MT_CHUNK;
    const float* p = a.ptr[t];
    const long long n = a.numel[t];
    float s = 0.f;
    const bool vec = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    for (int i = threadIdx.x * 4; i < MT_CHUNK; i += 1024) {
        const long long idx = base + i;
        if (vec && idx + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(p + idx);
            s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        } else {
            for (int j = 0; j < 4; ++j)
                if (idx + j < n) s += p[idx + j] * p[idx + j];
        }
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (double)s;
        pgroup[blockIdx.x] = a.group[t];
    }
}

// one block; every thread walks its partial sums ONCE and keeps one accumulator per group (each in the order the per-group loops of the
// first version took: ascending block index, stride 256), then the groups' tree reductions run side by side - the same sums, bit for bit,
// in a fifth of the time (35 -> ~10 us of a 1.7 ms one-sample step)
// A launch folds MT_FOLD_GROUPS groups (encoder + negative tokens + 14 layers); deeper configurations take one launch per 16 groups
// (g0 = the launch's first group), the last of which sums the total over ALL groups in group order from what the earlier ones wrote.
constexpr int MT_FOLD_GROUPS = 16;
__global__ __launch_bounds__(256) void mt_fold_kernel(const double* __restrict__ partial, const int* __restrict__ pgroup,
                                                     int n_blocks, int n_groups, int g0, int last, double* out) {
    __shared__ double red[MT_FOLD_GROUPS][256];
    double s[MT_FOLD_GROUPS];
#pragma unroll
    for (int g = 0; g < MT_FOLD_GROUPS; ++g) s[g] = 0.0;
#pragma unroll 8
    for (int i = threadIdx.x; i < n_blocks; i += 256) {
        const int pg = pgroup[i];
        const double v = partial[i];
#pragma unroll
        for (int g = 0; g < MT_FOLD_GROUPS; ++g) s[g] += pg == g0 + g ? v : 0.0;
    }
#pragma unroll
    for (int g = 0; g < MT_FOLD_GROUPS; ++g) red[g][threadIdx.x] = s[g];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
#pragma unroll
            for (int g = 0; g < MT_FOLD_GROUPS; ++g) red[g][threadIdx.x] += red[g][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int g = g0; g < n_groups && g < g0 + MT_FOLD_GROUPS; ++g) out[g] = red[g - g0][0];
        if (last) {
            double total = 0.0;  // out[n_groups] = sum over all groups (the clip kernel reads it)
            for (int g = 0; g < n_groups; ++g) total += g >= g0 ? red[g - g0][0] : out[g];
            out[n_groups] = total;
        }
    }
}

struct MtScaleArgs {
    float* ptr[MT_MAX];
    long long numel[MT_MAX];
    int first_block[MT_MAX + 1];
    int n;
};

// g *= min(1, max_norm / (sqrt(total_sq) + 1e-6))   (torch.nn.utils.clip_grad_norm_ semantics)
__global__ __launch_bounds__(256) void mt_clip_kernel(const MtScaleArgs a, const double* __restrict__ total_sq, float max_norm) {
    const double tn = sqrt(*total_sq);
    const float coef = (float)fmin(1.0, (double)max_norm / (tn + 1e-6));
    if (coef >= 1.f) return;
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long base = (long long)(blockIdx.x - a.first_block[lo]) * MT_CHUNK;
    float* p = a.ptr[lo];
    const long long n = a.numel[lo];
    for (int i = threadIdx.x; i < MT_CHUNK; i += 256)
        if (base + i < n) p[base + i] *= coef;
}


// ---- clip + AdamW in one multi-tensor pass (round 5) ---------------------------------------------------------------------------------
// train.py:121-125 at the reference's batch size is ~130 launches of 2-25 us; clipping (one pass over the 132 MB of gradients) and torch's
// fused AdamW (three launches) are six of them.  Here: ONE launch that scales the gradient by the clip coefficient (the decision taken in
// the kernel from the device-side norm, as mt_clip_kernel does; the scaled gradient is written back, so .grad holds what torch's path
// leaves there) and applies the AdamW update with torch's own arithmetic - torch/include/ATen/native/cuda/fused_adam_utils.cuh, adam_math
// with ADAM_MODE::ADAMW, no amsgrad / maximize / grad scaler: the scalars are doubles, the tensors floats, the expressions below are
// written with the same types and in the same order, so the updated weights and moments are bit-identical to torch.optim.AdamW(fused=True)
// (tests/test_gpu_backward.py::test_fused_clip_adamw_equals_torch).  The table of tensor pointers lives in device memory (84 tensors x
// four pointers exceed what a launch should carry as arguments).  The update's number comes from the optimizer's OWN device step tensors
// (torch increments first: this update is number *step + 1): every block reads its entry's counter, the block that finishes LAST (ticket
// counter) writes counter + 1 into every entry - so a torch optimizer.step() in between, or a load_state_dict, can never leave a stale
// host-side count behind.  `step` > 0 = an explicit number for callers without step tensors.
struct AdamEntry {
    float* p;
    float* g;
    float* m;
    float* v;
    float* step;  // torch's per-parameter step tensor (device float): read for the update's number, advanced by the last block
    long long numel;
    int first_block;
    int pad;
};
__global__ __launch_bounds__(256) void mt_clip_adamw_kernel(const AdamEntry* __restrict__ tab, int n, const double* __restrict__ total_sq, float max_norm,
                                                             double lr, double beta1, double beta2, double eps, double weight_decay, float step_arg, int write_back,
                                                             int* __restrict__ ticket) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AdamEntry e = tab[lo];
    // a float counts exactly up to 2^24 and stays there, as torch's counter does
    const float step = step_arg > 0.f ? step_arg : *e.step + 1.f;
    float coef = 1.f;
    if (max_norm > 0.f) {  // mt_clip_kernel's coefficient
        const double tn = sqrt(*total_sq);
        coef = (float)fmin(1.0, (double)max_norm / (tn + 1e-6));
    }
    const bool clip = coef < 1.f;
    // torch: bias_correction1 = 1 - pow(beta1, step), bias_correction2_sqrt = sqrt(1 - pow(beta2, step)), computed in double, used as floats
    const float bias_correction1 = (float)(1 - pow(beta1, (double)step));
    const float bias_correction2_sqrt = (float)sqrt(1 - pow(beta2, (double)step));
    const long long base = (long long)(blockIdx.x - e.first_block) * MT_CHUNK;
    for (int i = threadIdx.x * 4; i < MT_CHUNK; i += 1024) {
        const long long idx = base + i;
        if (idx >= e.numel) break;
        const int cnt = (int)((e.numel - idx) < 4 ? (e.numel - idx) : 4);
        float pr[4], gr[4], mr[4], vr[4];
        const bool vec = cnt == 4 && ((reinterpret_cast<uintptr_t>(e.p + idx) | reinterpret_cast<uintptr_t>(e.g + idx) | reinterpret_cast<uintptr_t>(e.m + idx) |
                                       reinterpret_cast<uintptr_t>(e.v + idx)) & 15) == 0;
        if (vec) {
            const float4 a = *reinterpret_cast<const float4*>(e.p + idx), b = *reinterpret_cast<const float4*>(e.g + idx);
            const float4 c = *reinterpret_cast<const float4*>(e.m + idx), d = *reinterpret_cast<const float4*>(e.v + idx);
            pr[0] = a.x; pr[1] = a.y; pr[2] = a.z; pr[3] = a.w; gr[0] = b.x; gr[1] = b.y; gr[2] = b.z; gr[3] = b.w;
            mr[0] = c.x; mr[1] = c.y; mr[2] = c.z; mr[3] = c.w; vr[0] = d.x; vr[1] = d.y; vr[2] = d.z; vr[3] = d.w;
        } else {
            for (int j = 0; j < cnt; ++j) { pr[j] = e.p[idx + j]; gr[j] = e.g[idx + j]; mr[j] = e.m[idx + j]; vr[j] = e.v[idx + j]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= cnt) break;
            float param = pr[j];
            float grad = gr[j];
            if (clip) grad *= coef;
            gr[j] = grad;
            float exp_avg = mr[j];
            float exp_avg_sq = vr[j];
            if (weight_decay != 0) param -= lr * weight_decay * param;
            exp_avg = beta1 * exp_avg + (1 - beta1) * grad;
            exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad * grad;
            const float step_size = lr / bias_correction1;
            float denom;
            denom = (sqrtf(exp_avg_sq) / bias_correction2_sqrt) + eps;
            param -= step_size * exp_avg / denom;
            pr[j] = param; mr[j] = exp_avg; vr[j] = exp_avg_sq;
        }
        if (vec) {
            *reinterpret_cast<float4*>(e.p + idx) = make_float4(pr[0], pr[1], pr[2], pr[3]);
            *reinterpret_cast<float4*>(e.m + idx) = make_float4(mr[0], mr[1], mr[2], mr[3]);
            *reinterpret_cast<float4*>(e.v + idx) = make_float4(vr[0], vr[1], vr[2], vr[3]);
            if (clip && write_back) *reinterpret_cast<float4*>(e.g + idx) = make_float4(gr[0], gr[1], gr[2], gr[3]);
        } else {
            for (int j = 0; j < cnt; ++j) {
                e.p[idx + j] = pr[j]; e.m[idx + j] = mr[j]; e.v[idx + j] = vr[j];
                if (clip && write_back) e.g[idx + j] = gr[j];
            }
        }
    }
    // every block has read its counter by now (its stores depend on it; the barrier below is behind every thread's use of it); the last
    // one to arrive advances all of them.  A relaxed agent-scope atomic and NO fence: nothing another block wrote is read here - only the
    // count matters - and a __threadfence() (L2 write-back + invalidate on gfx950) in each of the 8 052 blocks cost the one-sample step
    // 0.4 ms (1.53 -> 1.98 ms)
    __shared__ int is_last;
    __syncthreads();
    if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (is_last) {
        for (int i = threadIdx.x; i < n; i += 256) {
            float* sp = tab[i].step;
            if (sp) *sp = step_arg > 0.f ? step_arg : *sp + 1.f;
        }
        if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

size_t mt_sqnorm_scratch_bytes(int n, const long long* numel) {
    long long blocks = 0;
    for (int i = 0; i < n; ++i) blocks += (numel[i] + MT_CHUNK - 1) / MT_CHUNK;
    return (size_t)blocks * (sizeof(double) + sizeof(int)) + 64;
}

int launch_mt_sqnorm(const float* const* ptrs, const long long* numel, const int* group, int n, int n_groups, double* out,
                     void* scratch, size_t scratch_bytes, hipStream_t s) {
    SOLA_ARG(n > 0 && n <= MT_MAX && n_groups > 0, "grad_sqnorms: %d tensors (max %d)", n, MT_MAX);
    MtArgs a;
    a.n = n;
    int blocks = 0;
    double bytes = 0;
    for (int i = 0; i < n; ++i) {
        SOLA_ARG(ptrs[i] && numel[i] > 0 && group[i] >= 0 && group[i] < n_groups, "grad_sqnorms: bad tensor %d", i);
        a.ptr[i] = ptrs[i]; a.numel[i] = numel[i]; a.group[i] = group[i];
        a.first_block[i] = blocks;
        blocks += (int)((numel[i] + MT_CHUNK - 1) / MT_CHUNK);
        bytes += 4.0 * numel[i];
    }
    a.first_block[n] = blocks;
    const size_t need = (size_t)blocks * (sizeof(double) + sizeof(int)) + 64;
    if (scratch_bytes < need) {
        sola_set_error("grad_sqnorms: scratch %zu < %zu", scratch_bytes, need);
        return SOLA_ERR_WORKSPACE;
    }
    double* partial = static_cast<double*>(scratch);
    int* pgroup = reinterpret_cast<int*>(partial + blocks);
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, bytes);
    hipLaunchKernelGGL(mt_sqnorm_kernel, dim3(blocks), dim3(256), 0, s, a, partial, pgroup);
    SOLA_LAUNCH_CHECK();
    for (int g0 = 0; g0 < n_groups; g0 += MT_FOLD_GROUPS) {
        hipLaunchKernelGGL(mt_fold_kernel, dim3(1), dim3(256), 0, s, partial, pgroup, blocks, n_groups, g0, g0 + MT_FOLD_GROUPS >= n_groups ? 1 : 0, out);
        SOLA_LAUNCH_CHECK();
    }
    return SOLA_OK;
}

int launch_mt_clip(float* const* ptrs, const long long* numel, int n, const double* total_sq, float max_norm, hipStream_t s) {
    SOLA_ARG(n > 0 && n <= MT_MAX && total_sq, "grad_clip: bad arguments");
    MtScaleArgs a;
    a.n = n;
    int blocks = 0;
    double bytes = 0;
    for (int i = 0; i < n; ++i) {
        a.ptr[i] = ptrs[i]; a.numel[i] = numel[i];
        a.first_block[i] = blocks;
        blocks += (int)((numel[i] + MT_CHUNK - 1) / MT_CHUNK);
        bytes += 8.0 * numel[i];
    }
    a.first_block[n] = blocks;
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, bytes);
    hipLaunchKernelGGL(mt_clip_kernel, dim3(blocks), dim3(256), 0, s, a, total_sq, max_norm);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}


// tab_dev: n AdamEntry records in device memory (built by the caller: api.hip sola_adamw_bind); blocks = the table's total chunk count
size_t mt_adam_entry_bytes() { return sizeof(AdamEntry); }
void mt_adam_entry_fill(void* host_entry, float* p, float* g, float* m, float* v, float* step, long long numel, int first_block) {
    AdamEntry* e = static_cast<AdamEntry*>(host_entry);
    e->p = p; e->g = g; e->m = m; e->v = v; e->step = step; e->numel = numel; e->first_block = first_block; e->pad = 0;
}
int mt_adam_blocks(long long numel) { return (int)((numel + MT_CHUNK - 1) / MT_CHUNK); }
int launch_mt_clip_adamw(const void* tab_dev, int n, int blocks, double bytes, const double* total_sq, float max_norm, double lr, double beta1, double beta2,
                         double eps, double weight_decay, float step, int write_back, int* ticket, hipStream_t s) {
    SOLA_ARG(tab_dev && n > 0 && blocks > 0 && step >= 0.f && ticket && (max_norm <= 0.f || total_sq), "clip_adamw: bad arguments");
    SolaProfScope prof(SOLA_PROF_MISC, s, 0, bytes);
    hipLaunchKernelGGL(mt_clip_adamw_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const AdamEntry*>(tab_dev), n, total_sq, max_norm, lr, beta1, beta2, eps,
                       weight_decay, step, write_back, ticket);
    SOLA_LAUNCH_CHECK();
    return SOLA_OK;
}
