// One optimizer step's device work in ONE call (round 5): the reference's training loop body - forward, weighted BCE + alignment loss,
// backward, per-group gradient norms, clipping (train.py:62-125 at its batch size, configs/mevis/default.yaml:37) - enqueued from C++.
// The step is ~110 launches of 2-20 us; driven call by call from Python (autograd Function shells, ctypes) its host side cost 1.6-2.4 ms
// and WAS the wall on a slower host (421 samples/s on the driver's box against 530 here).  A hipGraph replay of the same
// launches is slower than the eager stream (2.06 against 1.83 ms, tools/train_one_graph_probe.py), so the launches stay stream launches
// and the host side becomes one call.  Same kernels, same arguments, same order as the Python path: gradients are bit-identical
// (tests/test_gpu_backward.py::test_train_step_call_equals_the_autograd_path).  The optimizer update stays the caller's (torch's fused
// AdamW reads the gradients where this call left them: the module's gradient arena).
#include <string>
#include <vector>

#include "ctx.h"

namespace {

__global__ __launch_bounds__(256) void vec_add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] += b[i];
}
__global__ void fill3_kernel(float* g, float x, float y, float z) {
    if (threadIdx.x == 0) { g[0] = x; g[1] = y; g[2] = z; }
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

struct StepLayout {
    size_t d_sm, d_st, d_neg, loss_scr, bwd_scr, g3, sq_scr, total;
};
StepLayout step_layout(const SolaCtx* c, int B, int N, size_t sq_scratch) {
    const size_t D = (size_t)c->cfg.lang_token_dim, nn = (size_t)c->cfg.n_negative;
    StepLayout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
    l.d_sm = take((size_t)B * N * 4);
    l.d_st = take((size_t)B * N * D * 4);
    l.d_neg = take(nn * D * 4);
    l.loss_scr = take((size_t)B * N * 3 * 4);
    l.bwd_scr = take(((((size_t)B * N * nn + 63) & ~(size_t)63) + (size_t)B * nn * D) * 4);
    l.g3 = take(3 * 4);
    l.sq_scr = take(sq_scratch);
    l.total = off;
    return l;
}

}  // namespace

extern "C" int sola_train_step_bind(SolaCtx* c, const char* const* names, const int32_t* group, int n, int n_groups) {
    SOLA_ARG(c && names && group && n > 0 && n <= 128 && n_groups > 0, "train_step_bind: bad argument");
    try {
        SolaCtx::StepBinding b;
        b.n_groups = n_groups;
        for (int i = 0; i < n; ++i) {
            auto it = c->index.find(names[i]);
            SOLA_ARG(it != c->index.end(), "train_step_bind: unknown parameter '%s'", names[i]);
            SOLA_ARG(group[i] >= 0 && group[i] < n_groups, "train_step_bind: group %d of '%s' out of range", (int)group[i], names[i]);
            b.widx.push_back(it->second);
            b.group.push_back(group[i]);
            b.numel.push_back(c->weights[it->second].numel);
        }
        b.sq_scratch = mt_sqnorm_scratch_bytes(n, b.numel.data());
        c->step = std::move(b);
        return SOLA_OK;
    } catch (const std::exception& e) {
        sola_set_error("train_step_bind: %s", e.what());
        return SOLA_ERR_ARG;
    }
}

extern "C" size_t sola_train_step_workspace_bytes(const SolaCtx* c, int B, int N) {
    if (!c || B <= 0 || N <= 0) return 0;
    if (c->step.widx.empty()) return 0;
    return step_layout(c, B, N, c->step.sq_scratch).total;
}

extern "C" int sola_train_step(SolaCtx* c, const float* obj, const float* lang, int B, int N, int T, int L, const float* labels, const float* pos,
                               float positive_weight, float temperature, float alignment_weight, float max_grad_norm, float* score_map,
                               float* score_tokens, float* loss3, double* grad_sq, void* train_ws, size_t train_ws_bytes, void* bwd_ws,
                               size_t bwd_ws_bytes, void* step_ws, size_t step_ws_bytes, void* stream_) {
    SOLA_ARG(c && obj && lang && labels && pos && score_map && score_tokens && loss3 && grad_sq && train_ws && bwd_ws && step_ws, "train_step: null argument");
    try {
        SOLA_ARG(!c->step.widx.empty(), "train_step: sola_train_step_bind has not been called on this context");
        const SolaCtx::StepBinding& b = c->step;
        const StepLayout lay = step_layout(c, B, N, b.sq_scratch);
        if (step_ws_bytes < lay.total) {
            sola_set_error("train_step: step workspace %zu bytes < required %zu", step_ws_bytes, lay.total);
            return SOLA_ERR_WORKSPACE;
        }
        hipStream_t s = as_stream(stream_);
        char* w = static_cast<char*>(step_ws);
        float* d_sm = reinterpret_cast<float*>(w + lay.d_sm);
        float* d_st = reinterpret_cast<float*>(w + lay.d_st);
        float* d_neg = reinterpret_cast<float*>(w + lay.d_neg);
        float* g3 = reinterpret_cast<float*>(w + lay.g3);
        const int D = c->cfg.lang_token_dim, n_neg = c->cfg.n_negative;
        const float* neg = ctx_weight(c, "negative_token.weight");
        float* g_neg = ctx_grad(c, "negative_token.weight");
        SOLA_ARG(neg && g_neg, "train_step: negative_token.weight / its gradient are not bound");
        // forward (activations kept in train_ws), the three losses on the negative tokens themselves (train.py:92 repeats them per sample:
        // the shared-table form of sola_loss), d(total) / d(score_map, score_tokens, negative tokens)
        SOLA_TRY(sola_forward_impl(c, obj, lang, B, N, T, L, score_map, score_tokens, train_ws, train_ws_bytes, s, true));
        // one sample (the reference's batch): the [1, n_neg, D] table of train.py:92 - per-sample form, as the autograd path passes it
        const int64_t stride = B == 1 ? (int64_t)n_neg * D : 0;
        SOLA_TRY(sola_loss(score_map, score_tokens, labels, pos, neg, stride, B, N, D, n_neg, positive_weight, temperature, alignment_weight, loss3,
                           nullptr, w + lay.loss_scr, (size_t)B * N * 3 * 4, stream_));
        hipLaunchKernelGGL(fill3_kernel, dim3(1), dim3(64), 0, s, g3, 1.f, 0.f, 0.f);  // d(loss3[0]) = (1, 0, 0): train.py:116 backpropagates the total
        SOLA_LAUNCH_CHECK();
        SOLA_TRY(sola_loss_backward(score_map, score_tokens, labels, pos, neg, stride, B, N, D, n_neg, positive_weight, temperature, alignment_weight, g3,
                                    d_sm, d_st, d_neg, w + lay.bwd_scr, lay.g3 - lay.bwd_scr, stream_));
        SOLA_TRY(sola_backward(c, d_sm, d_st, train_ws, bwd_ws, bwd_ws_bytes, stream_));
        // the negative tokens collect a second contribution, straight from the alignment loss (autograd adds the two)
        {
            const long long n = (long long)n_neg * D;
            hipLaunchKernelGGL(vec_add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g_neg, d_neg, n);
            SOLA_LAUNCH_CHECK();
        }
        // module/module.py:164-199 + train.py:121-122: per-group sums of squares (+ total) on the device, the clip decision in the kernel
        const int n = (int)b.widx.size();
        const float* gp[128];
        float* gq[128];
        for (int i = 0; i < n; ++i) {
            gq[i] = c->weights[b.widx[i]].grad;
            SOLA_ARG(gq[i], "train_step: the gradient of '%s' is not bound (sola_set_grad)", c->weights[b.widx[i]].name.c_str());
            gp[i] = gq[i];
        }
        SOLA_TRY(launch_mt_sqnorm(gp, b.numel.data(), b.group.data(), n, b.n_groups, grad_sq, w + lay.sq_scr, b.sq_scratch, s));
        if (max_grad_norm > 0.f) SOLA_TRY(launch_mt_clip(gq, b.numel.data(), n, grad_sq + b.n_groups, max_grad_norm, s));
        return SOLA_OK;
    } catch (const std::exception& e) {
        sola_set_error("train_step: %s", e.what());
        return SOLA_ERR_ARG;
    }
}


// ---- clip + AdamW in one launch (optim.hip: mt_clip_adamw_kernel) --------------------------------------------------------------------
// bind: the optimizer's state tensors by parameter name (torch.optim.AdamW keeps exp_avg / exp_avg_sq / step per parameter; with fused=True
// the step is a device float).  The parameter and gradient pointers are the context's own bindings (sola_set_weight / sola_set_grad): the
// update writes the CALLER's parameter storage in place.  Bind again whenever one of the pointers changes: sola_set_weight / sola_set_grad
// with a NEW pointer drop the table (sola_adamw_step then fails with "bind again" instead of writing through a stale pointer).  The table is
// rewritten behind a device synchronisation, so an update still in flight on any stream never sees half of it.
extern "C" int sola_adamw_bind(SolaCtx* c, const char* const* names, void* const* exp_avg, void* const* exp_avg_sq, void* const* step, int n) {
    SOLA_ARG(c && names && exp_avg && exp_avg_sq && n > 0 && n <= 128, "adamw_bind: bad argument");
    try {
        const size_t eb = mt_adam_entry_bytes();
        std::vector<char> host((size_t)n * eb);
        int blocks = 0;
        double bytes = 0;
        for (int i = 0; i < n; ++i) {
            auto it = c->index.find(names[i]);
            SOLA_ARG(it != c->index.end(), "adamw_bind: unknown parameter '%s'", names[i]);
            const Weight& w = c->weights[it->second];
            SOLA_ARG(w.ptr && w.grad && exp_avg[i] && exp_avg_sq[i], "adamw_bind: '%s' needs its weight, gradient and both moments bound", names[i]);
            mt_adam_entry_fill(host.data() + (size_t)i * eb, const_cast<float*>(w.ptr), w.grad, static_cast<float*>(exp_avg[i]), static_cast<float*>(exp_avg_sq[i]),
                               step ? static_cast<float*>(step[i]) : nullptr, w.numel, blocks);
            blocks += mt_adam_blocks(w.numel);
            bytes += 28.0 * (double)w.numel;  // p, g, m, v in; p, m, v out
        }
        c->adam_n = 0;
        SOLA_HIP(hipDeviceSynchronize());  // rare (once per optimizer / pointer change): no update may be reading the table
        if (c->adam_tab && c->adam_cap < n) { (void)hipFree(c->adam_tab); c->adam_tab = nullptr; }
        if (!c->adam_tab) { SOLA_HIP(hipMalloc(&c->adam_tab, (size_t)n * eb)); c->adam_cap = n; }
        if (!c->adam_ticket) SOLA_HIP(hipMalloc(&c->adam_ticket, sizeof(int)));
        SOLA_HIP(hipMemset(c->adam_ticket, 0, sizeof(int)));
        SOLA_HIP(hipMemcpy(c->adam_tab, host.data(), (size_t)n * eb, hipMemcpyHostToDevice));
        bool has_steps = step != nullptr;
        for (int i = 0; has_steps && i < n; ++i) has_steps = step[i] != nullptr;
        c->adam_has_steps = has_steps;
        c->adam_n = n; c->adam_blocks = blocks; c->adam_bytes = bytes;
        return SOLA_OK;
    } catch (const std::exception& e) {
        sola_set_error("adamw_bind: %s", e.what());
        return SOLA_ERR_ARG;
    }
}

// step = 0: the update's number is read from the bound step tensors on the device (number = counter + 1, counters advanced by the kernel -
// what torch's own step() does, so the two can be mixed freely on one optimizer); step >= 1: an explicit number (1 for the first), written
// into the step tensors if there are any.  dev_total_sq = the gradients' total sum of squares on the device (sola_train_step's
// dev_grad_sq + n_groups) when max_grad_norm > 0, else ignored.  Invalidate derived weight copies afterwards (sola_weights_changed).
// write_back_grads: 1 = an active clip leaves the scaled gradients in the gradient tensors (what clip_grad_norm_ leaves in .grad); 0 = they
// keep the unclipped values (one eighth less traffic: train.py never reads .grad behind optimizer.step()) - parameters and moments the same
extern "C" int sola_adamw_step(SolaCtx* c, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, const double* dev_total_sq,
                               float max_grad_norm, int write_back_grads, void* stream_) {
    SOLA_ARG(c && c->adam_tab && c->adam_n > 0, "adamw_step: no optimizer bound to this context (sola_adamw_bind; bind again after sola_set_weight / sola_set_grad moved a pointer)");
    SOLA_ARG(step >= 0, "adamw_step: step %lld", (long long)step);
    SOLA_ARG(step >= 1 || c->adam_has_steps, "adamw_step: step 0 reads the optimizer's device step tensors, and none were bound");
    if (step > (1 << 24)) step = 1 << 24;  // a float counter stays at 2^24, as torch's does
    SOLA_TRY(launch_mt_clip_adamw(c->adam_tab, c->adam_n, c->adam_blocks, c->adam_bytes, dev_total_sq, max_grad_norm, lr, beta1, beta2, eps, weight_decay, (float)step,
                                  write_back_grads, c->adam_ticket, as_stream(stream_)));
    c->ws_dirty = true;
    c->lin16_dirty = true;
    return SOLA_OK;
}
