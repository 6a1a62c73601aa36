"""Per-stage operators over libsola_hip.so, on torch CUDA tensors (device memory + stream plumbing only).

Each function is one HIP kernel family of the track-selection path; shapes follow the channels-last
convention of the library (include/sola_hip.h).  No CPU path exists: CPU tensors raise.
"""
from __future__ import annotations

import math

import torch

from . import _lib
from ._lib import check, current_stream, lib, ptr, require_cuda


def _f32c(t):
    if t.dtype != torch.float32:
        raise _lib.SolaError(f"expected float32, got {t.dtype}")
    return t.contiguous()


def ws_standardize(weight):
    """module/ws.py:9-13.  weight [cout, cin, k] -> standardised [cout, k*cin] (k-major GEMM layout)."""
    require_cuda(weight)
    w = _f32c(weight)
    cout, cin, k = w.shape
    out = torch.empty((cout, k * cin), device=w.device, dtype=torch.float32)
    check(lib().sola_ws_standardize(ptr(w), cout, cin, k, ptr(out), current_stream(w.device)), "sola_ws_standardize")
    return out


def gemm_nt(a, w, bias=None, residual=None):
    """a [M,K] @ w[N,K]^T + bias (+ residual [M,N])  (F.linear, tools/attention.py:63-65,73)."""
    require_cuda(a, w, bias, residual)
    a, w = _f32c(a), _f32c(w)
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    bias = None if bias is None else _f32c(bias)
    residual = None if residual is None else _f32c(residual)
    check(lib().sola_gemm_nt(ptr(a), K, ptr(w), ptr(bias), ptr(residual), N, ptr(out), N, M, N, K,
                             current_stream(a.device)), "sola_gemm_nt")
    return out


def conv1d_cl(x, w_std, bias, k, stride, pad):
    """Channels-last conv along T (module/ws.py:14-22): x [R,T,cin], w_std [cout,k*cin] -> [R,T_out,cout]."""
    require_cuda(x, w_std, bias)
    x, w_std = _f32c(x), _f32c(w_std)
    R, T, cin = x.shape
    cout = w_std.shape[0]
    t_out = (T + 2 * pad - k) // stride + 1
    y = torch.empty((R, t_out, cout), device=x.device, dtype=torch.float32)
    check(lib().sola_conv1d_cl(ptr(x), ptr(w_std), ptr(None if bias is None else _f32c(bias)), ptr(y), R, T, cin, cout,
                               k, stride, pad, current_stream(x.device)), "sola_conv1d_cl")
    return y


def group_norm(x, gamma, beta, groups, n_inst, inner, outer_stride, inner_stride, tok_stride, ntok, eps=1e-5,
               leaky_slope=None, pe=None):
    """nn.GroupNorm over token sets of an [rows, C] matrix (see sola_group_norm in the header).
    Returns y, or (y, y + pe[inst % inner]) when ``pe`` is given."""
    require_cuda(x, gamma, beta, pe)
    x = _f32c(x)
    C_ = x.shape[-1]
    y = torch.empty_like(x)
    y2 = torch.empty_like(x) if pe is not None else None
    check(lib().sola_group_norm(ptr(x), ptr(y), ptr(y2), ptr(None if pe is None else _f32c(pe)), ptr(_f32c(gamma)),
                                ptr(_f32c(beta)), n_inst, inner, outer_stride, inner_stride, tok_stride, ntok, C_,
                                groups, eps, 0.0 if leaky_slope is None else leaky_slope,
                                0 if leaky_slope is None else 1, current_stream(x.device)), "sola_group_norm")
    return y if pe is None else (y, y2)


def attention(q, k, v, G, H, Sq, Sk, inner, q_addr, k_addr, scale=None):
    """softmax(q k^T * scale) v over G groups x H heads; q,k,v are [rows, H*dh] matrices and
    ``q_addr``/``k_addr`` = (outer, inner_stride, row_stride) in rows (tools/attention.py:66-72)."""
    require_cuda(q, k, v)
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    D = q.shape[-1]
    dh = D // H
    o = torch.zeros_like(q)
    scale = 1.0 / math.sqrt(dh) if scale is None else scale
    check(lib().sola_attention(ptr(q), q.shape[-1], ptr(k), k.shape[-1], ptr(v), v.shape[-1], ptr(o), D, G, H, dh, Sq, Sk,
                               inner, q_addr[0], q_addr[1], q_addr[2], k_addr[0], k_addr[1], k_addr[2], scale,
                               current_stream(q.device)), "sola_attention")
    return o


def pos_encoding(gauss, t_len, max_temporal_length):
    """module/module.py:112-128 -> [t_len, D]."""
    require_cuda(gauss)
    g = _f32c(gauss).reshape(-1)
    D = g.numel() * 2
    pe = torch.empty((t_len, D), device=g.device, dtype=torch.float32)
    check(lib().sola_pos_encoding(ptr(g), D, t_len, max_temporal_length, ptr(pe), current_stream(g.device)), "sola_pos_encoding")
    return pe


def select(score_map, threshold=0.5):
    """inference.py:59-60: (sigmoid(score), sigmoid(score) > threshold as float)."""
    require_cuda(score_map)
    s = _f32c(score_map)
    prob = torch.empty_like(s)
    pred = torch.empty_like(s)
    check(lib().sola_select(ptr(s), s.numel(), threshold, ptr(prob), ptr(pred), current_stream(s.device)), "sola_select")
    return prob, pred
