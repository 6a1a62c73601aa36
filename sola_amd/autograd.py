"""Differentiable (training) path: ``torch.autograd.Function`` shells around sola_forward_train / sola_backward and
sola_loss / sola_loss_backward.  PyTorch only carries the tensors and the graph edges; every gradient is computed by
the HIP kernels of libsola_hip.so (backward.hip).  This is what ``loss.backward()`` at train.py:116-117 runs."""
from __future__ import annotations

import torch

from ._lib import SolaError, check, current_stream, lib, ptr


class _TrackSelection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, object_tokens, lang_tokens, *params):
        score_map, score_tokens = module._forward_train_impl(object_tokens, lang_tokens)
        ctx.module = module
        ctx.n_params = len(params)
        ctx.gen = module._train_generation
        return score_map, score_tokens

    @staticmethod
    def backward(ctx, d_score_map, d_score_tokens):
        m = ctx.module
        if ctx.gen != m._train_generation:
            raise SolaError("sola_amd: backward through a stale forward (the module ran another training forward since)")
        grads = m._backward_impl(d_score_map, d_score_tokens)
        return (None, None, None, *grads)


def track_selection_forward(module, object_tokens, lang_tokens):
    params = module._param_list()
    return _TrackSelection.apply(module, object_tokens, lang_tokens, *params)


class _TrackSelectionRagged(torch.autograd.Function):
    """The ragged training step: sola_forward_train_ragged / sola_backward_ragged over lists of per-sample tensors."""

    @staticmethod
    def forward(ctx, module, n_samples, *tensors):
        objs, langs = list(tensors[:n_samples]), list(tensors[n_samples:2 * n_samples])
        score_map, score_tokens = module._forward_train_ragged_impl(objs, langs)
        ctx.module = module
        ctx.n_inputs = 2 * n_samples
        ctx.gen = module._train_generation
        return score_map, score_tokens

    @staticmethod
    def backward(ctx, d_score_map, d_score_tokens):
        m = ctx.module
        if ctx.gen != m._train_generation:
            raise SolaError("sola_amd: backward through a stale forward (the module ran another training forward since)")
        grads = m._backward_impl(d_score_map, d_score_tokens)
        return (None, None, *([None] * ctx.n_inputs), *grads)


def track_selection_forward_ragged(module, object_tokens, lang_tokens):
    params = module._param_list()
    return _TrackSelectionRagged.apply(module, len(lang_tokens), *object_tokens, *lang_tokens, *params)


class _Losses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score_map, score_tokens, labels, pos_tokens, neg_tokens, positive_weight, temperature, alignment_weight):
        from .loss import _loss_forward

        loss3, _ = _loss_forward(score_map, score_tokens, labels, pos_tokens, neg_tokens, positive_weight, temperature,
                                 alignment_weight, False)
        ctx.save_for_backward(score_map, score_tokens, labels, pos_tokens, neg_tokens)
        ctx.hyper = (float(positive_weight), float(temperature), float(alignment_weight))
        return loss3

    @staticmethod
    def backward(ctx, g3):
        score_map, score_tokens, labels, pos_tokens, neg_tokens = ctx.saved_tensors
        pw, temp, aw = ctx.hyper
        B, N = score_map.shape
        D = score_tokens.shape[-1]
        f = lambda t: t.detach().to(torch.float32).contiguous()
        sm, st, lb, ps, ng, g = f(score_map), f(score_tokens), f(labels), f(pos_tokens), f(neg_tokens), f(g3)
        shared = ng.dim() == 2
        n_neg = ng.shape[0] if shared else ng.shape[1]
        stride = 0 if shared else n_neg * D
        dev = sm.device
        d_sm = torch.empty_like(sm)
        d_st = torch.empty_like(st)
        d_neg = torch.empty_like(ng) if ctx.needs_input_grad[4] else None
        n_scratch = ((B * N * n_neg + 63) // 64) * 64 + (B * n_neg * D if (shared and d_neg is not None) else 0)
        scratch = torch.empty(n_scratch, device=dev, dtype=torch.float32)
        check(lib().sola_loss_backward(ptr(sm), ptr(st), ptr(lb), ptr(ps), ptr(ng), stride, B, N, D, n_neg, pw, temp, aw,
                                       ptr(g), ptr(d_sm), ptr(d_st), ptr(d_neg), ptr(scratch), scratch.numel() * 4,
                                       current_stream(dev)), "sola_loss_backward")
        return d_sm, d_st, None, None, d_neg, None, None, None


class _LossesRagged(torch.autograd.Function):
    """sola_loss_ragged / sola_loss_backward_ragged: [S, 3] per-sample {total, bce, alignment}, each a mean over the sample's own
    tracks (train.py:98-113 at the reference's batch size of 1)."""

    @staticmethod
    def forward(ctx, score_map, score_tokens, labels, pos_tokens, neg_tokens, track_offsets, counts, positive_weight, temperature,
                alignment_weight):
        from .loss import _loss_forward_ragged

        loss3, _ = _loss_forward_ragged(score_map, score_tokens, labels, pos_tokens, neg_tokens, track_offsets, counts,
                                        positive_weight, temperature, alignment_weight, False)
        ctx.save_for_backward(score_map, score_tokens, labels, pos_tokens, neg_tokens, track_offsets)
        ctx.hyper = (float(positive_weight), float(temperature), float(alignment_weight), list(counts))
        return loss3

    @staticmethod
    def backward(ctx, g3):
        score_map, score_tokens, labels, pos_tokens, neg_tokens, track_offsets = ctx.saved_tensors
        pw, temp, aw, counts = ctx.hyper
        S, total = len(counts), int(score_map.numel())
        D = score_tokens.shape[-1]
        f = lambda t: t.detach().to(torch.float32).contiguous()
        sm, st, lb, ps, ng, g = f(score_map), f(score_tokens), f(labels), f(pos_tokens), f(neg_tokens), f(g3)
        shared = ng.dim() == 2
        n_neg = ng.shape[0] if shared else ng.shape[1]
        stride = 0 if shared else n_neg * D
        dev = sm.device
        d_sm = torch.empty_like(sm)
        d_st = torch.empty_like(st)
        d_neg = torch.empty_like(ng) if ctx.needs_input_grad[4] else None
        n_scratch = ((total * n_neg + 63) // 64) * 64 + (S * n_neg * D if (shared and d_neg is not None) else 0)
        scratch = torch.empty(n_scratch, device=dev, dtype=torch.float32)
        check(lib().sola_loss_backward_ragged(ptr(sm), ptr(st), ptr(lb), ptr(ps), ptr(ng), stride, S, ptr(track_offsets),
                                              max(counts), total, D, n_neg, pw, temp, aw, ptr(g), ptr(d_sm), ptr(d_st),
                                              ptr(d_neg), ptr(scratch), scratch.numel() * 4, current_stream(dev)),
              "sola_loss_backward_ragged")
        return d_sm, d_st, None, None, d_neg, None, None, None, None, None
