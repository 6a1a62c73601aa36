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

__global__ __launch_bounds__(256) void mt_fold_kernel(const double* __restrict__ partial, const int* __restrict__ pgroup,
                                                     int n_blocks, int n_groups, double* __restrict__ out) {
    __shared__ double red[256];
    double total = 0.0;  // out[n_groups] = sum over all groups (the clip kernel reads it)
    for (int g = 0; g < n_groups; ++g) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n_blocks; i += 256)
            if (pgroup[i] == g) s += partial[i];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            out[g] = red[0];
            total += red[0];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n_groups] = total;
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
    hipLaunchKernelGGL(mt_fold_kernel, dim3(1), dim3(256), 0, s, partial, pgroup, blocks, n_groups, out);
    SOLA_LAUNCH_CHECK();
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
