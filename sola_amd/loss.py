"""Drop-in for ``tools/loss.py`` (AlignmentLoss) and the loss assembly of ``train.py:98-113`` on libsola_hip.so."""
from __future__ import annotations

import torch
import torch.nn as nn

from ._lib import SolaError, check, current_stream, lib, ptr, require_cuda


def _loss_forward(score_map, score_tokens, labels, pos_tokens, neg_tokens, positive_weight, temperature, alignment_weight,
                  return_argmax):
    require_cuda(score_map, score_tokens, labels, pos_tokens, neg_tokens)
    B, N = score_map.shape
    D = score_tokens.shape[-1]
    f = lambda t: t.detach().to(torch.float32).contiguous()
    sm, st, lb, ps, ng = f(score_map), f(score_tokens), f(labels), f(pos_tokens), f(neg_tokens)
    if ps.numel() != B * D:
        raise SolaError("pos_tokens must be [B,1,D] (n_pos must be 1, tools/loss.py:24)")
    if ng.dim() == 2:
        n_neg, stride = ng.shape[0], 0
    else:
        if ng.shape[0] != B:
            raise SolaError("neg_tokens batch dimension mismatch")
        n_neg, stride = ng.shape[1], ng.shape[1] * D
    dev = sm.device
    loss3 = torch.empty(3, device=dev, dtype=torch.float32)
    argmax = torch.empty((B, N), device=dev, dtype=torch.int32) if return_argmax else None
    scratch = torch.empty(B * N * 3, device=dev, dtype=torch.float32)
    check(lib().sola_loss(ptr(sm), ptr(st), ptr(lb), ptr(ps), ptr(ng), stride, B, N, D, n_neg, float(positive_weight),
                          float(temperature), float(alignment_weight), ptr(loss3), ptr(argmax), ptr(scratch),
                          scratch.numel() * 4, current_stream(dev)), "sola_loss")
    return loss3, argmax


def track_selection_losses(score_map, score_tokens, labels, pos_tokens, neg_tokens, positive_weight=1.5,
                           temperature=0.07, alignment_weight=0.3, return_argmax=False):
    """One fused evaluation of train.py:98-113: weighted BCE on the logits, AlignmentLoss, and their sum.

    score_map [B,N], score_tokens [B,N,D], labels [B,N], pos_tokens [B,1,D]; neg_tokens [B,n_neg,D] or a
    shared [n_neg,D] table (what train.py:92 builds by repeating ``negative_token.weight``).
    Returns a float32 tensor ``[total, bce, alignment]`` on the device (no host sync); differentiable with respect to
    score_map, score_tokens and neg_tokens (HIP backward, sola_loss_backward)."""
    needs_grad = torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in (score_map, score_tokens, neg_tokens))
    if needs_grad and return_argmax:
        raise SolaError("track_selection_losses: return_argmax=True is an evaluation option; call it under torch.no_grad() (or on "
                        "detached inputs), and call again without it for the differentiable loss")
    if needs_grad:
        from .autograd import _Losses

        return _Losses.apply(score_map, score_tokens, labels, pos_tokens, neg_tokens, positive_weight, temperature,
                             alignment_weight)
    loss3, argmax = _loss_forward(score_map, score_tokens, labels, pos_tokens, neg_tokens, positive_weight, temperature,
                                  alignment_weight, return_argmax)
    return (loss3, argmax) if return_argmax else loss3


class AlignmentLoss(nn.Module):
    """tools/loss.py:4-58.  ``temperature`` stays a Parameter for interface parity; the reference never
    optimises it (train.py:44-49), so its value is read on the host at call time and it receives no gradient."""

    def __init__(self, positive_weight: float = 1.0, temperature: float = 0.07) -> None:
        super().__init__()
        self.positive_weight = positive_weight
        self.temperature = nn.Parameter(torch.tensor(temperature))

    def forward(self, object_tokens, labels, pos_tokens, neg_tokens):
        assert pos_tokens.shape[1] == 1, "n_pos must be 1"
        zeros = torch.zeros(labels.shape, device=object_tokens.device, dtype=torch.float32)
        loss3 = track_selection_losses(zeros, object_tokens, labels, pos_tokens, neg_tokens,
                                       positive_weight=self.positive_weight, temperature=float(self.temperature.detach()),
                                       alignment_weight=0.0)
        return loss3[2]


def weighted_bce_with_logits(score_logits, labels, positive_weight=1.5):
    """train.py:98-104: F.binary_cross_entropy_with_logits(input, target, weight=where(labels>0, pw, 1)), mean."""
    B, N = score_logits.shape
    dev = score_logits.device
    D = 4
    z = torch.zeros((B, N, D), device=dev, dtype=torch.float32)
    loss3 = track_selection_losses(score_logits, z, labels, torch.zeros((B, 1, D), device=dev), torch.zeros((1, D), device=dev),
                                   positive_weight=positive_weight, temperature=0.0, alignment_weight=0.0)
    return loss3[1]


def _loss_forward_ragged(score_map, score_tokens, labels, pos_tokens, neg_tokens, track_offsets, counts, positive_weight,
                         temperature, alignment_weight, return_argmax):
    require_cuda(score_map, score_tokens, labels, pos_tokens, neg_tokens, track_offsets)
    S = len(counts)
    total = int(score_map.numel())
    D = score_tokens.shape[-1]
    if track_offsets.dtype != torch.int32 or track_offsets.numel() != S + 1 or sum(counts) != total:
        raise SolaError("track_offsets must be int32 [S+1] and counts must sum to the number of tracks")
    f = lambda t: t.detach().to(torch.float32).contiguous()
    sm, st, lb, ps, ng = f(score_map), f(score_tokens), f(labels), f(pos_tokens), f(neg_tokens)
    if ps.numel() != S * D or lb.numel() != total:
        raise SolaError("pos_tokens must be [S,D] and labels [sum N_i]")
    if ng.dim() == 2:
        n_neg, stride = ng.shape[0], 0
    else:
        if ng.shape[0] != S:
            raise SolaError("neg_tokens batch dimension mismatch")
        n_neg, stride = ng.shape[1], ng.shape[1] * D
    dev = sm.device
    loss3 = torch.empty((S, 3), device=dev, dtype=torch.float32)
    argmax = torch.empty(total, device=dev, dtype=torch.int32) if return_argmax else None
    scratch = torch.empty(total * 3, device=dev, dtype=torch.float32)
    check(lib().sola_loss_ragged(ptr(sm), ptr(st), ptr(lb), ptr(ps), ptr(ng), stride, S, ptr(track_offsets.contiguous()),
                                 max(counts), total, D, n_neg, float(positive_weight), float(temperature),
                                 float(alignment_weight), ptr(loss3), ptr(argmax), ptr(scratch), scratch.numel() * 4,
                                 current_stream(dev)), "sola_loss_ragged")
    return loss3, argmax


def track_selection_losses_ragged(score_map, score_tokens, labels, pos_tokens, neg_tokens, track_offsets, counts,
                                  positive_weight=1.5, temperature=0.07, alignment_weight=0.3, return_argmax=False):
    """``track_selection_losses`` for the flat outputs of ``forward_ragged``: every sample gets its OWN means over its own
    tracks - what train.py:98-113 / evaluator.py:88-112 compute at the reference's batch size of 1 (sola_loss_ragged).

    score_map [sum N_i], score_tokens [sum N_i, D], labels [sum N_i]; pos_tokens [S, D] (or [S,1,D]); neg_tokens a shared
    [n_neg, D] table or [S, n_neg, D]; track_offsets int32 [S+1] on the device; counts = python list of N_i.
    Returns float32 [S, 3] = {total, bce, alignment} per sample (and int32 [sum N_i] hardest-negative indices).
    Differentiable with respect to score_map, score_tokens and neg_tokens (sola_loss_backward_ragged): ``loss[:, 0].mean()`` is
    the batch objective whose gradient is the average of the reference's per-sample (batch-size-1) gradients."""
    needs_grad = torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in (score_map, score_tokens, neg_tokens))
    if needs_grad and return_argmax:
        # the hardest-negative indices come from the non-differentiable evaluation: silently handing back a loss without a graph
        #  would train without the alignment loss's gradient
        raise SolaError("track_selection_losses_ragged: return_argmax=True is an evaluation option; call it under torch.no_grad() "
                        "(or on detached inputs), and call again without it for the differentiable loss")
    if needs_grad:
        from .autograd import _LossesRagged

        return _LossesRagged.apply(score_map, score_tokens, labels, pos_tokens, neg_tokens, track_offsets.contiguous(), list(counts),
                                   positive_weight, temperature, alignment_weight)
    with torch.no_grad():
        loss3, argmax = _loss_forward_ragged(score_map, score_tokens, labels, pos_tokens, neg_tokens, track_offsets, counts,
                                             positive_weight, temperature, alignment_weight, return_argmax)
    return (loss3, argmax) if return_argmax else loss3
