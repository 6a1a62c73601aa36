"""CPU ORACLE for the SOLA track-selection hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, in plain PyTorch-CPU tensor algebra, of the algorithm the
reference runs for this path.  It is the *checker* for the HIP implementation: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product
package (``sola_amd``) never imports it and has no CPU fallback.

Parity status: PINNED.  ``tests/golden/gen_golden.py`` imports the real reference
(``/root/reference/module/module.py``, ``tools/loss.py``, ``train.py:98-113`` loss assembly) in the
authoring container, runs it on seeded inputs and commits the outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function below against those vectors (the reference
itself ships no tests or golden vectors).

Third-party arithmetic: the reference's math lives in PyTorch library ops (pinned torch==2.6.0 in
requirements.txt:36): F.conv1d, F.linear, F.scaled_dot_product_attention, nn.GroupNorm,
F.binary_cross_entropy_with_logits.  They are restated here from their published definitions
(im2col contraction, softmax(QK^T/sqrt(d))V, biased-variance group normalisation, the stable
BCE-with-logits formula) in a channels-last layout, so that the oracle is an independent derivation
rather than a call into the same operators.

Layout convention: activations are ``[B, N, T, C]`` (channels last) throughout.
"""
from __future__ import annotations

import math

import numpy as np
import torch

NUM_HEADS = 8  # module/module.py:13-15
LEAKY_SLOPE = 0.01  # nn.LeakyReLU() default, module/module.py:77
GN_EPS = 1e-5  # nn.GroupNorm default eps
WS_EPS = 1e-5  # module/ws.py:11


def _t(x, dtype):
    if isinstance(x, np.ndarray):
        x = torch.tensor(x)
    if x.requires_grad:  # tests differentiate through the oracle with torch autograd
        return x.to(device="cpu", dtype=dtype)
    return x.detach().to(device="cpu", dtype=dtype)


def to_torch_state(sd, dtype=torch.float32):
    return {k: _t(v, dtype) for k, v in sd.items()}


# --------------------------------------------------------------------------------------------
# a1: weight-standardised conv1d (module/ws.py:8-22)
# --------------------------------------------------------------------------------------------
def standardize_weight(w):
    """w [cout, cin, k] -> (w - mean) / (std_unbiased + 1e-5), statistics per output channel over
    (cin, k) (module/ws.py:9-13; torch.std default is the n-1 estimator; eps is added to the std)."""
    cout = w.shape[0]
    flat = w.reshape(cout, -1)
    n = flat.shape[1]
    mean = flat.sum(dim=1, keepdim=True) / n
    cen = flat - mean
    var = (cen * cen).sum(dim=1, keepdim=True) / (n - 1)
    return (cen / (var.sqrt() + WS_EPS)).reshape(w.shape)


def conv1d_cl(x, w, bias, stride, pad):
    """Channels-last 1-D convolution along T as an im2col contraction (F.conv1d semantics, ws.py:14-22).
    x [R, T, cin], w [cout, cin, k] -> [R, T_out, cout]."""
    R, T, cin = x.shape
    cout, _, k = w.shape
    t_out = (T + 2 * pad - k) // stride + 1
    xp = torch.zeros((R, T + 2 * pad, cin), dtype=x.dtype)
    xp[:, pad:pad + T] = x
    cols = []
    for kk in range(k):
        cols.append(xp[:, kk:kk + stride * (t_out - 1) + 1:stride, :])  # [R, t_out, cin]
    col = torch.cat(cols, dim=2)  # [R, t_out, k*cin]   (k-major, cin-minor)
    wmat = w.permute(0, 2, 1).reshape(cout, k * cin)  # same (k, cin) ordering
    return col @ wmat.t() + bias


# --------------------------------------------------------------------------------------------
# GroupNorm with statistics over (channels-in-group x tokens) (nn.GroupNorm, module.py:76,19,34,43,49)
# --------------------------------------------------------------------------------------------
def group_norm_tokens(x, gamma, beta, groups):
    """x [I, S, C]: for every instance i and group g, normalise over the S tokens and the C/groups
    channels of the group with the biased variance, then apply the per-channel affine."""
    I, S, C = x.shape
    cg = C // groups
    xg = x.reshape(I, S, groups, cg)
    cnt = S * cg
    mean = xg.sum(dim=(1, 3), keepdim=True) / cnt
    cen = xg - mean
    var = (cen * cen).sum(dim=(1, 3), keepdim=True) / cnt
    y = cen / torch.sqrt(var + GN_EPS)
    return y.reshape(I, S, C) * gamma + beta


def leaky_relu(x):
    return torch.where(x >= 0, x, x * LEAKY_SLOPE)


# --------------------------------------------------------------------------------------------
# a2: short-term motion encoder (module/module.py:74-96,137-140), eval mode (dropout off)
# --------------------------------------------------------------------------------------------
_CONV_IDX = (0, 4, 8, 12, 16, 20)
_NORM_IDX = (1, 5, 9, 13, 17)
_CONV_GEOM = ((3, 2, 1), (3, 2, 1), (3, 2, 1), (3, 1, 1), (3, 1, 1), (1, 1, 0))  # (k, stride, pad)


def encoder(sd, cfg, obj, taps=None):
    """obj [B, N, T, d] -> [B, N, T', D]."""
    B, N, T, d = obj.shape
    x = obj.reshape(B * N, T, d)
    for li in range(6):
        w = sd[f"short_motion_encoder.{_CONV_IDX[li]}.weight"]
        b = sd[f"short_motion_encoder.{_CONV_IDX[li]}.bias"]
        _k, s, p = _CONV_GEOM[li]
        x = conv1d_cl(x, standardize_weight(w), b, s, p)
        if taps is not None:
            taps[f"conv{li}"] = x.reshape(B, N, x.shape[1], x.shape[2])
        if li < 5:
            g = sd[f"short_motion_encoder.{_NORM_IDX[li]}.weight"]
            be = sd[f"short_motion_encoder.{_NORM_IDX[li]}.bias"]
            x = leaky_relu(group_norm_tokens(x, g, be, cfg["n_groups"]))
    return x.reshape(B, N, x.shape[1], x.shape[2])


# --------------------------------------------------------------------------------------------
# a3: Fourier temporal positional encoding (module/module.py:112-128)
# --------------------------------------------------------------------------------------------
def positional_encoding(sd, cfg, t_len, dtype):
    """[t_len, D] table: [sin(2*pi*(t/max_len)*G), cos(...)] with G the [1, D/2] random buffer."""
    G = sd["positional_encoding_gaussian_matrix"]
    t = torch.arange(t_len).to(dtype).reshape(-1, 1) / cfg["max_temporal_length"]
    arg = (t @ G) * (2 * np.pi)
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


# --------------------------------------------------------------------------------------------
# a4: multi-head attention (tools/attention.py:54-74)
# --------------------------------------------------------------------------------------------
def attention(sd, prefix, q_in, k_in, v_in):
    """q_in [G, Sq, D], k_in/v_in [G, Sk, D] -> [G, Sq, D]; no mask, scale 1/sqrt(dh), eval mode."""
    def lin(x, name):
        return x @ sd[f"{prefix}.{name}.weight"].t() + sd[f"{prefix}.{name}.bias"]

    G, Sq, D = q_in.shape
    Sk = k_in.shape[1]
    dh = D // NUM_HEADS
    q = lin(q_in, "q_proj").reshape(G, Sq, NUM_HEADS, dh).permute(0, 2, 1, 3)
    k = lin(k_in, "k_proj").reshape(G, Sk, NUM_HEADS, dh).permute(0, 2, 1, 3)
    v = lin(v_in, "v_proj").reshape(G, Sk, NUM_HEADS, dh).permute(0, 2, 1, 3)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    s = s - s.max(dim=-1, keepdim=True).values
    p = torch.exp(s)
    p = p / p.sum(dim=-1, keepdim=True)
    o = (p @ v).permute(0, 2, 1, 3).reshape(G, Sq, D)
    return lin(o, "out_proj")


# --------------------------------------------------------------------------------------------
# a5: one object-language alignment layer (module/module.py:22-52)
# --------------------------------------------------------------------------------------------
def align_layer(sd, cfg, layer, x, pe, lang, taps=None):
    """x [B, N, T', D], pe [T', D], lang [B, W, D] -> x'."""
    B, N, Tp, D = x.shape
    p = f"object_lang_align_layers.{layer}"
    ng = cfg["n_groups_module"]

    def gn(idx, z):
        return group_norm_tokens(z, sd[f"{p}.norm.{idx}.weight"], sd[f"{p}.norm.{idx}.bias"], ng)

    # (i) inter-object attention over the N tracks of each (b, t'); GN statistics over 128ch x N (:31-35)
    z = x.permute(0, 2, 1, 3).reshape(B * Tp, N, D)
    z = gn(0, z + attention(sd, f"{p}.obj_attn", z, z, z))
    x = z.reshape(B, Tp, N, D).permute(0, 2, 1, 3)
    if taps is not None:
        taps[f"l{layer}_obj"] = x.clone()
    # (ii) motion attention over T' per track; q,k carry the PE, v does not (:38-43)
    z = x.reshape(B * N, Tp, D)
    zp = z + pe
    z = gn(1, z + attention(sd, f"{p}.motion_attn", zp, zp, z))
    if taps is not None:
        taps[f"l{layer}_motion"] = z.reshape(B, N, Tp, D).clone()
    # (iii) object -> language cross attention over all N*T' tokens of the sample (:46-50)
    z = z.reshape(B, N * Tp, D)
    z = gn(2, z + attention(sd, f"{p}.object2lang_attn", z, lang, lang))
    x = z.reshape(B, N, Tp, D)
    if taps is not None:
        taps[f"l{layer}_o2l"] = x.clone()
    return x


# --------------------------------------------------------------------------------------------
# a6: whole forward (module/module.py:130-162)
# --------------------------------------------------------------------------------------------
def forward(sd, cfg, object_tokens, lang_tokens, dtype=torch.float32, taps=None):
    """-> (score_map [B, N], score_tokens [B, N, D]).  ``sd`` is a dict of arrays/tensors."""
    sd = to_torch_state(sd, dtype)
    obj = _t(object_tokens, dtype)
    lang = _t(lang_tokens, dtype)
    B = obj.shape[0]
    x = encoder(sd, cfg, obj, taps)
    if taps is not None:
        taps["encoder"] = x.clone()
    pe = positional_encoding(sd, cfg, x.shape[2], dtype)
    if taps is not None:
        taps["pe"] = pe.clone()
    neg = sd["negative_token.weight"].unsqueeze(0).expand(B, -1, -1)
    lang = torch.cat([lang, neg], dim=1)  # [B, W, D]  (:146-147)
    for layer in range(cfg["n_layers"]):
        x = align_layer(sd, cfg, layer, x, pe, lang, taps)
    # score head (:152-160): mean over the W text/negative tokens of the dot products
    logits = torch.einsum("bntd,bwd->bntw", x, lang).mean(dim=-1)  # [B, N, T']
    logits = logits - logits.max(dim=-1, keepdim=True).values
    a = torch.exp(logits)
    a = a / a.sum(dim=-1, keepdim=True)
    score_tokens = (x * a.unsqueeze(-1)).sum(dim=2)  # [B, N, D]
    score_map = torch.einsum("bnd,bwd->bnw", score_tokens, lang).mean(dim=-1)
    return score_map, score_tokens


# --------------------------------------------------------------------------------------------
# a8 / a9: losses (train.py:98-113, tools/loss.py:14-58)
# --------------------------------------------------------------------------------------------
def bce_with_logits(x, y, weight=None):
    """Elementwise max(x,0) - x*y + log1p(exp(-|x|)), optional weight (F.binary_cross_entropy_with_logits)."""
    v = torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))
    return v if weight is None else v * weight


def losses(score_map, score_tokens, labels, pos_tokens, neg_tokens, positive_weight=1.5,
           temperature=0.07, alignment_weight=0.3, dtype=torch.float32):
    """-> dict(total, bce, align, neg_argmax).  Weighted BCE on the logits (train.py:98-104), the
    alignment loss with hardest-negative masking (tools/loss.py:29-56) and their sum (train.py:113)."""
    x = _t(score_map, dtype)
    tok = _t(score_tokens, dtype)
    y = _t(labels, dtype)
    pos = _t(pos_tokens, dtype)
    neg = _t(neg_tokens, dtype)
    w = torch.where(y > 0, torch.full_like(y, positive_weight), torch.ones_like(y))
    bce = bce_with_logits(x, y, w).mean()
    s = math.exp(temperature)  # exp(nn.Parameter(temperature)), never optimised (train.py:44-49)
    pos_logits = torch.einsum("bnd,bmd->bnm", tok, pos) * s  # [B, N, 1]
    neg_logits = torch.einsum("bnd,bmd->bnm", tok, neg) * s  # [B, N, n_neg]
    idx = neg_logits.argmax(dim=-1)  # first maximum
    onehot = torch.zeros_like(neg_logits)
    onehot.scatter_(-1, idx.unsqueeze(-1), 1.0)
    neg_labels = (1 - y).unsqueeze(-1) * onehot
    pos_loss = bce_with_logits(pos_logits, y.unsqueeze(-1)).mean()
    neg_loss = bce_with_logits(neg_logits, neg_labels).mean()
    align = positive_weight * pos_loss + neg_loss
    return {"total": bce + alignment_weight * align, "bce": bce, "align": align, "neg_argmax": idx}


# --------------------------------------------------------------------------------------------
# a10: selection (inference.py:59-60)
# --------------------------------------------------------------------------------------------
def select(score_map, threshold=0.5):
    p = torch.sigmoid(_t(score_map, torch.float32))
    return (p > threshold).to(torch.float32)


# --------------------------------------------------------------------------------------------
# a7: grouped gradient norms (module/module.py:164-199)
# --------------------------------------------------------------------------------------------
def grad_norm_dict(grads, n_layers):
    """grads: dict key -> array.  Returns the five-entry dict of the reference."""
    def sq(keys):
        return float(sum((np.asarray(grads[k], dtype=np.float64) ** 2).sum() for k in keys))

    enc = [k for k in grads if k.startswith("short_motion_encoder.")]
    out = {"short_motion_encoder": sq(enc)}
    total = out["short_motion_encoder"]
    for li in range(n_layers):
        v = sq([k for k in grads if k.startswith(f"object_lang_align_layers.{li}.")])
        out[f"scmola_layer_{li}"] = v
        total += v
    out["negative_token"] = sq([k for k in grads if k.startswith("negative_token.")])
    total += out["negative_token"]
    out["total_grad_norm"] = total
    return {k: math.sqrt(v) for k, v in out.items()}
