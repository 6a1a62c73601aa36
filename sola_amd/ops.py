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


def set_stage_dropout(p=0.0, seed=0):
    """Dropout applied by group_norm / attention (and their backward) below; p = 0 turns it off."""
    check(lib().sola_set_stage_dropout(float(p), int(seed)), "sola_set_stage_dropout")


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


def cast_sp16(x, scale=1.0):
    """f32 [rows, K] -> split-f16 rows (a float32-typed container of the same shape holding [8 x f16 hi | 8 x f16 lo]
    per 8 values, hi + lo = scale * x to 22 bits).  ``scale`` must be a power of two."""
    require_cuda(x)
    x = _f32c(x)
    rows, K = x.shape
    out = torch.empty_like(x)
    check(lib().sola_cast_sp16(ptr(x), K, ptr(out), K, rows, K, float(scale), current_stream(x.device)), "sola_cast_sp16")
    return out


def decode_sp16(x_sp):
    """Inverse of ``cast_sp16`` (host-side view arithmetic, for tests): returns hi + lo as float32."""
    rows, K = x_sp.shape
    h = x_sp.contiguous().view(torch.float16).reshape(rows, K // 8, 2, 8).to(torch.float32)
    return (h[:, :, 0, :] + h[:, :, 1, :]).reshape(rows, K)


def cast_sp16_auto(x):
    """f32 [rows, K] -> (split-f16 rows scaled by a data-dependent power of two, scal) where scal[1] is the inverse scale
    on the device (pass it as ``out_scale_dev`` of ``gemm_nt_split``) and scal[0] = max|x|."""
    require_cuda(x)
    x = _f32c(x)
    rows, K = x.shape
    out = torch.empty_like(x)
    scal = torch.empty(2, device=x.device, dtype=torch.float32)
    check(lib().sola_cast_sp16_auto(ptr(x), K, ptr(out), K, rows, K, ptr(scal), current_stream(x.device)), "sola_cast_sp16_auto")
    return out, scal


def gemm_nt_split(a_sp, w_sp, bias=None, residual=None, residual_is_split=False, out_scale=1.0, out_split=False, out_scale_dev=None):
    """out_scale * (A W^T) + bias (+ residual) on split-f16 operands (three f16 MFMAs per product, f32 accumulate);
    ``out_split`` writes the result as split-f16 pairs (decode with ``decode_sp16``)."""
    require_cuda(a_sp, w_sp, bias, residual)
    a_sp, w_sp = _f32c(a_sp), _f32c(w_sp)
    M, K = a_sp.shape
    N = w_sp.shape[0]
    out = torch.empty((M, N), device=a_sp.device, dtype=torch.float32)
    check(lib().sola_gemm_nt_split_scaled(ptr(a_sp), K, ptr(w_sp), ptr(None if bias is None else _f32c(bias)),
                                          ptr(None if residual is None else _f32c(residual)), N, 1 if residual_is_split else 0,
                                          ptr(out), N, 1 if out_split else 0, M, N, K, float(out_scale), ptr(out_scale_dev),
                                          current_stream(a_sp.device)), "sola_gemm_nt_split_scaled")
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


def attention(q, k, v, G, H, Sq, Sk, inner, q_addr, k_addr, scale=None, return_lse=False):
    """softmax(q k^T * scale) v over G groups x H heads; q,k,v are [rows, H*dh] matrices and
    ``q_addr``/``k_addr`` = (outer, inner_stride, row_stride) in rows (tools/attention.py:66-72)."""
    require_cuda(q, k, v)
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    D = q.shape[-1]
    dh = D // H
    o = torch.zeros_like(q)
    scale = 1.0 / math.sqrt(dh) if scale is None else scale
    lse = torch.zeros((q.shape[0], H), device=q.device, dtype=torch.float32) if return_lse else None
    check(lib().sola_attention(ptr(q), q.shape[-1], ptr(k), k.shape[-1], ptr(v), v.shape[-1], ptr(o), D, G, H, dh, Sq, Sk,
                               inner, q_addr[0], q_addr[1], q_addr[2], k_addr[0], k_addr[1], k_addr[2], scale, ptr(lse),
                               current_stream(q.device)), "sola_attention")
    return (o, lse) if return_lse else o


def attention_split(q_sp, k_sp, v_sp, G, H, Sq, Sk, inner, q_addr, k_addr, scale=None, out_split=False):
    """``attention`` on split-f16 q, k, v (``cast_sp16`` rows): three f16 MFMAs per product, f32 accumulation."""
    require_cuda(q_sp, k_sp, v_sp)
    q_sp, k_sp, v_sp = _f32c(q_sp), _f32c(k_sp), _f32c(v_sp)
    D = q_sp.shape[-1]
    dh = D // H
    o = torch.zeros_like(q_sp)
    scale = 1.0 / math.sqrt(dh) if scale is None else scale
    check(lib().sola_attention_split(ptr(q_sp), D, ptr(k_sp), k_sp.shape[-1], ptr(v_sp), v_sp.shape[-1], ptr(o), D,
                                     1 if out_split else 0, G, H, dh, Sq, Sk, inner, q_addr[0], q_addr[1], q_addr[2],
                                     k_addr[0], k_addr[1], k_addr[2], scale, None, current_stream(q_sp.device)),
          "sola_attention_split")
    return o


def attention_backward(q, k, v, o, dout, lse, G, H, Sq, Sk, inner, q_addr, k_addr, scale=None):
    """Backward of ``attention``: returns (dq, dk, dv) with the layouts of q, k, v."""
    require_cuda(q, k, v, o, dout, lse)
    q, k, v, o, dout, lse = (_f32c(t) for t in (q, k, v, o, dout, lse))
    D = q.shape[-1]
    dh = D // H
    scale = 1.0 / math.sqrt(dh) if scale is None else scale
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    dvec = torch.zeros((q.shape[0], H), device=q.device, dtype=torch.float32)
    # scratch of the one-pass kernel's chunked launches (long query ranges against <= 64 keys; sola_hip.h)
    n_scr = int(lib().sola_attention_backward_scratch_floats(q.shape[0], G, H, Sk))
    scr = torch.empty(max(n_scr, 1), device=q.device, dtype=torch.float32)
    check(lib().sola_attention_backward_ws(ptr(q), q.shape[-1], ptr(k), k.shape[-1], ptr(v), v.shape[-1], ptr(o), ptr(dout), D,
                                           ptr(lse), ptr(dq), ptr(dk), ptr(dv), ptr(dvec), G, H, dh, Sq, Sk, inner,
                                           q_addr[0], q_addr[1], q_addr[2], k_addr[0], k_addr[1], k_addr[2], scale,
                                           q.shape[0], ptr(scr) if n_scr else None, n_scr, current_stream(q.device)), "sola_attention_backward_ws")
    return dq, dk, dv


def gemm_tn(a, b, want_bias_grad=False):
    """a [M,N]^T @ b [M,K] -> [N,K] (weight gradient); optionally also the column sums of a (bias gradient)."""
    require_cuda(a, b)
    a, b = _f32c(a), _f32c(b)
    M, N = a.shape
    K = b.shape[1]
    out = torch.empty((N, K), device=a.device, dtype=torch.float32)
    bias = torch.empty((N,), device=a.device, dtype=torch.float32) if want_bias_grad else None
    nb = lib().sola_gemm_tn_scratch_bytes(M, N, K)
    scratch = torch.empty(nb, device=a.device, dtype=torch.uint8)
    check(lib().sola_gemm_tn(ptr(a), N, ptr(b), K, ptr(out), ptr(bias), M, N, K, ptr(scratch), nb, current_stream(a.device)),
          "sola_gemm_tn")
    return (out, bias) if want_bias_grad else out


def gemm_tn_split(a, b):
    """a [M,N]^T @ b [M,K] -> [N,K] on the split-f16 MFMA path (transposing casts + split-K NT GEMM)."""
    require_cuda(a, b)
    a, b = _f32c(a), _f32c(b)
    M, N = a.shape
    K = b.shape[1]
    out = torch.empty((N, K), device=a.device, dtype=torch.float32)
    nb = lib().sola_gemm_tn_split_scratch_bytes(M, N, K)
    scratch = torch.empty(nb, device=a.device, dtype=torch.uint8)
    check(lib().sola_gemm_tn_split(ptr(a), N, ptr(b), K, ptr(out), M, N, K, ptr(scratch), nb, current_stream(a.device)),
          "sola_gemm_tn_split")
    return out


def gemm_tn_f16(a, b, bf16=False):
    """a [M,N]^T @ b [M,K] -> [N,K] on plain f16 / bf16 operands (one MFMA per product; for N, K multiples of 256 without any
    transposed copy: row-major casts + transposing LDS reads)."""
    require_cuda(a, b)
    a, b = _f32c(a), _f32c(b)
    M, N = a.shape
    K = b.shape[1]
    out = torch.empty((N, K), device=a.device, dtype=torch.float32)
    nb = lib().sola_gemm_tn_split_scratch_bytes(M, N, K)
    scratch = torch.empty(nb, device=a.device, dtype=torch.uint8)
    check(lib().sola_gemm_tn_f16(ptr(a), N, ptr(b), K, ptr(out), M, N, K, 2 if bf16 else 1, ptr(scratch), nb, current_stream(a.device)),
          "sola_gemm_tn_f16")
    return out


def conv1d_cl_wgrad_f16(x, dy, k, stride, pad, bf16=False):
    """Weight gradient of ``conv1d_cl`` on f16 / bf16 operands: x [R,T,cin], dy [R,T_out,cout] -> dw_std [cout, k*cin]."""
    require_cuda(x, dy)
    x, dy = _f32c(x), _f32c(dy)
    R, T, cin = x.shape
    cout = dy.shape[2]
    t_out = (T + 2 * pad - k) // stride + 1
    assert dy.shape[1] == t_out
    dw = torch.empty((cout, k * cin), device=x.device, dtype=torch.float32)
    nb = lib().sola_gemm_tn_split_scratch_bytes(R * t_out, cout, k * cin)
    scratch = torch.empty(nb, device=x.device, dtype=torch.uint8)
    check(lib().sola_conv1d_cl_wgrad_f16(ptr(x), ptr(dy), ptr(dw), R, T, cin, cout, k, stride, pad, 2 if bf16 else 1, ptr(scratch), nb,
                                         current_stream(x.device)), "sola_conv1d_cl_wgrad_f16")
    return dw


def ws_backward(weight, dwstd):
    """Backward of ``ws_standardize``: weight [cout,cin,k], dwstd [cout,k*cin] -> dweight [cout,cin,k]."""
    require_cuda(weight, dwstd)
    w, g = _f32c(weight), _f32c(dwstd)
    cout, cin, k = w.shape
    dw = torch.empty_like(w)
    check(lib().sola_ws_backward(ptr(w), ptr(g), cout, cin, k, ptr(dw), current_stream(w.device)), "sola_ws_backward")
    return dw


def conv1d_cl_backward(x, w_std, dy, k, stride, pad, need_dx=True, split=False):
    """Backward of ``conv1d_cl``: returns (dx or None, dw_std [cout,k*cin], dbias [cout]); ``split`` = the split-f16 path."""
    require_cuda(x, w_std, dy)
    x, w_std, dy = _f32c(x), _f32c(w_std), _f32c(dy)
    R, T, cin = x.shape
    cout = w_std.shape[0]
    t_out = (T + 2 * pad - k) // stride + 1
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty_like(w_std)
    db = torch.empty((cout,), device=x.device, dtype=torch.float32)
    if split:
        nb = lib().sola_conv1d_cl_backward_split_scratch_bytes(R, T, cin, cout, k, stride, pad)
        scratch = torch.empty(max(nb, 1), device=x.device, dtype=torch.uint8)
        check(lib().sola_conv1d_cl_backward_split(ptr(x), ptr(w_std), ptr(dy), ptr(dx), ptr(dw), ptr(db), R, T, cin, cout, k,
                                                  stride, pad, ptr(scratch), nb, current_stream(x.device)), "sola_conv1d_cl_backward_split")
        return dx, dw, db
    nb = max(lib().sola_gemm_tn_scratch_bytes(R * t_out, cout, k * cin), 4 * cout * k * cin)
    scratch = torch.empty(nb, device=x.device, dtype=torch.uint8)
    check(lib().sola_conv1d_cl_backward(ptr(x), ptr(w_std), ptr(dy), ptr(dx), ptr(dw), ptr(db), R, T, cin, cout, k, stride,
                                        pad, ptr(scratch), nb, current_stream(x.device)), "sola_conv1d_cl_backward")
    return dx, dw, db


def group_norm_backward(x, dy, gamma, beta, groups, n_inst, inner, outer_stride, inner_stride, tok_stride, ntok, eps=1e-5,
                        leaky_slope=None, dy2=None):
    """Backward of ``group_norm``: returns (dx, dgamma, dbeta); ``dy2`` is the gradient of the y+pe side output."""
    require_cuda(x, dy, gamma, beta, dy2)
    x, dy = _f32c(x), _f32c(dy)
    C_ = x.shape[-1]
    dx = torch.empty_like(x)
    dg = torch.empty((C_,), device=x.device, dtype=torch.float32)
    db = torch.empty((C_,), device=x.device, dtype=torch.float32)
    nb = 2 * n_inst * C_ * 4
    scratch = torch.empty(nb, device=x.device, dtype=torch.uint8)
    check(lib().sola_group_norm_backward(ptr(x), ptr(dy), ptr(None if dy2 is None else _f32c(dy2)), ptr(_f32c(gamma)),
                                         ptr(_f32c(beta)), ptr(dx), ptr(dg), ptr(db), n_inst, inner, outer_stride,
                                         inner_stride, tok_stride, ntok, C_, groups, eps,
                                         0.0 if leaky_slope is None else leaky_slope, 0 if leaky_slope is None else 1,
                                         ptr(scratch), nb, current_stream(x.device)), "sola_group_norm_backward")
    return dx, dg, db


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
