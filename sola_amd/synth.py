"""Deterministic synthetic weights and inputs for the track-selection path.

There is no dataset, no RoBERTa checkpoint and no network on either box, so parity
tests, golden-vector generation and ``bench.py`` all draw their tensors from the
formulas below. Everything comes from ``numpy.random.Generator(PCG64(seed))`` in a
fixed order, so the authoring container (where the reference is imported to make the
golden vectors) and the GPU box regenerate bit-identical float32 arrays.

State-dict key order and shapes follow the reference module
(module/module.py:74-110, tools/attention.py:26-29): encoder convs/GroupNorms, then per
layer obj_attn/motion_attn/object2lang_attn {q,k,v,out}_proj and norm.{0,1,2}, the
``positional_encoding_gaussian_matrix`` buffer and ``negative_token.weight``.
"""
from __future__ import annotations

import numpy as np

DEFAULT_MODEL_CFG = {
    # configs/mevis/default.yaml:3-13
    "object_token_dim": 256,
    "lang_token_dim": 1024,
    "n_layers": 2,
    "max_temporal_length": 100,
    "n_negative": 32,
    "dropout_p": 0.2,
    "norm_type": "group",
    "n_groups": 8,
    "n_groups_module": 8,
}

# A reduced configuration with the same structure (8 heads x 16) used for fast parity cases.
SMALL_MODEL_CFG = {
    "object_token_dim": 32,
    "lang_token_dim": 128,
    "n_layers": 2,
    "max_temporal_length": 100,
    "n_negative": 4,
    "dropout_p": 0.2,
    "norm_type": "group",
    "n_groups": 8,
    "n_groups_module": 8,
}

NUM_HEADS = 8  # hard-coded in module/module.py:13-15


def encoder_spec(cfg):
    """(cin, cout, k, stride, pad) of the six weight-standardised convs (module/module.py:75-95)."""
    d, h, D = cfg["object_token_dim"], cfg["object_token_dim"] * 2, cfg["lang_token_dim"]
    return [
        (d, h, 3, 2, 1),
        (h, h, 3, 2, 1),
        (h, h, 3, 2, 1),
        (h, D, 3, 1, 1),
        (D, D, 3, 1, 1),
        (D, D, 1, 1, 0),
    ]


def state_dict_spec(cfg):
    """Ordered list of (key, shape, kind) for every tensor of the reference state_dict."""
    D = cfg["lang_token_dim"]
    spec = []
    conv_idx = [0, 4, 8, 12, 16, 20]
    norm_idx = [1, 5, 9, 13, 17]
    for li, (cin, cout, k, _s, _p) in enumerate(encoder_spec(cfg)):
        spec.append((f"short_motion_encoder.{conv_idx[li]}.weight", (cout, cin, k), ("uniform", cin * k)))
        spec.append((f"short_motion_encoder.{conv_idx[li]}.bias", (cout,), ("uniform", cin * k)))
        if li < 5:
            spec.append((f"short_motion_encoder.{norm_idx[li]}.weight", (cout,), ("gamma", 0)))
            spec.append((f"short_motion_encoder.{norm_idx[li]}.bias", (cout,), ("beta", 0)))
    for layer in range(cfg["n_layers"]):
        p = f"object_lang_align_layers.{layer}"
        for attn in ("obj_attn", "motion_attn", "object2lang_attn"):
            for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
                spec.append((f"{p}.{attn}.{proj}.weight", (D, D), ("uniform", D)))
                spec.append((f"{p}.{attn}.{proj}.bias", (D,), ("uniform", D)))
        for j in range(3):
            spec.append((f"{p}.norm.{j}.weight", (D,), ("gamma", 0)))
            spec.append((f"{p}.norm.{j}.bias", (D,), ("beta", 0)))
    spec.append(("positional_encoding_gaussian_matrix", (1, D // 2), ("normal", 0)))
    spec.append(("negative_token.weight", (cfg["n_negative"], D), ("normal", 0)))
    return spec


def make_state_dict(cfg, seed=42):
    """float32 numpy state_dict.

    conv / linear weights and biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the scale of torch's
    default init), GroupNorm gamma = 1 + 0.1*N(0,1), beta = 0.1*N(0,1) (so the affine terms are
    exercised), the Fourier matrix and the negative tokens ~ N(0,1).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for key, shape, (kind, fan_in) in state_dict_spec(cfg):
        if kind == "uniform":
            bound = 1.0 / np.sqrt(float(fan_in))
            arr = rng.uniform(-bound, bound, size=shape)
        elif kind == "gamma":
            arr = 1.0 + 0.1 * rng.standard_normal(size=shape)
        elif kind == "beta":
            arr = 0.1 * rng.standard_normal(size=shape)
        else:
            arr = rng.standard_normal(size=shape)
        out[key] = np.ascontiguousarray(arr.astype(np.float32))
    return out


# Range cases of the split-f16 mode (tests/golden/range_golden.npz): weights away from the default-init scale and inputs
# away from N(0,1).  Every variant is a formula on top of make_state_dict, so both boxes regenerate it.
WEIGHT_VARIANTS = ("base", "lin_x2", "lin_x8", "lin_div64", "lin_outliers", "conv_x100", "gamma_div256", "gamma_x300", "gamma_x3000")
# (object-token scale, text-token scale).  The encoder's first GroupNorm removes the object-token scale, so any value is a
# well-posed case; the text tokens feed the object->language softmax directly, which saturates into an arg-max beyond a scale
# of ~16 - there the reference's own fp32 result is 1e-2 away from exact arithmetic and parity is not defined (see "cond").
RANGE_INPUT_SCALES = ((1e-5, 1.0), (1e-3, 1.0), (1e-1, 1.0), (10.0, 1.0), (1e3, 1.0), (1.0, 1e-5), (1.0, 1e-3), (1.0, 16.0),
                      (1e-4, 1e-2), (1e3, 4.0))


def make_state_dict_variant(cfg, seed=42, variant="base"):
    sd = make_state_dict(cfg, seed)
    if variant == "base":
        return sd
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    for key in sd:
        is_lin = "_proj.weight" in key
        is_conv = key.startswith("short_motion_encoder") and key.endswith(".weight") and sd[key].ndim == 3
        is_norm = (".norm." in key) or (key.startswith("short_motion_encoder") and sd[key].ndim == 1 and
                                        int(key.split(".")[1]) in (1, 5, 9, 13, 17))
        if variant == "lin_x2" and is_lin:
            sd[key] = sd[key] * np.float32(2.0)
        elif variant == "lin_x8" and is_lin:  # attention scores x64: saturated softmax, the reference itself is ill-conditioned here
            sd[key] = sd[key] * np.float32(8.0)
        elif variant == "lin_div64" and is_lin:
            sd[key] = sd[key] * np.float32(1.0 / 64.0)
        elif variant == "lin_outliers" and is_lin:  # 16 entries per matrix at 64x the init bound
            flat = sd[key].reshape(-1)
            idx = rng.choice(flat.size, size=16, replace=False)
            flat[idx] = np.where(rng.uniform(size=16) < 0.5, -2.0, 2.0).astype(np.float32)
        elif variant == "conv_x100" and is_conv:  # weight standardisation removes the scale (module/ws.py:9-13)
            sd[key] = sd[key] * np.float32(100.0)
        elif variant == "gamma_div256" and is_norm:  # GroupNorm outputs of rms ~ 0.004: below what the fixed activation scale covers
            sd[key] = sd[key] * np.float32(1.0 / 256.0)
        elif variant == "gamma_x300" and is_norm:
            sd[key] = sd[key] * np.float32(300.0)
        elif variant == "gamma_x3000" and is_norm:
            sd[key] = sd[key] * np.float32(3000.0)
    return {k: np.ascontiguousarray(v.astype(np.float32)) for k, v in sd.items()}


def make_inputs(cfg, B, N, T, L, seed=0, pos_rate=0.2):
    """Synthetic sample batch: object tokens and text tokens ~ N(0,1), labels ~ Bernoulli,
    pos_tokens = mean over the L text tokens (what train.py:86-90 yields for an unpadded batch)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d, D = cfg["object_token_dim"], cfg["lang_token_dim"]
    obj = rng.standard_normal(size=(B, N, T, d)).astype(np.float32)
    lang = rng.standard_normal(size=(B, L, D)).astype(np.float32)
    labels = (rng.uniform(size=(B, N)) < pos_rate).astype(np.float32)
    pos = lang.astype(np.float64).mean(axis=1, keepdims=True).astype(np.float32)
    return {"object_tokens": obj, "lang_tokens": lang, "labels": labels, "pos_tokens": pos}


def make_ragged_samples(cfg, n_samples, seed=2024, device=None, n_range=(8, 80), t_range=(20, 200), l_range=(4, 24), pos_rate=0.2):
    """A MeViS-like mix of variable-shape samples (every real sample has its own N tracks, T frames, L text tokens:
    dataloader.py:119-163, 187-199): N ~ U[8,80], T ~ U[20,200], L ~ U[4,24], seeded.  Returns a list of dicts of torch
    tensors {obj [N,T,d], lang [L,D], labels [N], pos [D]} (pos = mean of the text tokens, train.py:86-90)."""
    import torch

    rng = np.random.Generator(np.random.PCG64(seed))
    d, D = cfg["object_token_dim"], cfg["lang_token_dim"]
    out = []
    for _ in range(n_samples):
        N, T, L = (int(rng.integers(lo, hi + 1)) for lo, hi in (n_range, t_range, l_range))
        obj = rng.standard_normal(size=(N, T, d)).astype(np.float32)
        lang = rng.standard_normal(size=(L, D)).astype(np.float32)
        labels = (rng.uniform(size=N) < pos_rate).astype(np.float32)
        pos = lang.astype(np.float64).mean(axis=0).astype(np.float32)
        smp = {"obj": torch.from_numpy(obj), "lang": torch.from_numpy(lang), "labels": torch.from_numpy(labels), "pos": torch.from_numpy(pos)}
        if device is not None:
            smp = {k: v.to(device) for k, v in smp.items()}
        out.append(smp)
    return out


def make_ragged_infer_batches(cfg, n_samples=128, seed=2024):
    """The two ragged INFERENCE batches of bench.py's ragged leg (and of tests/test_gpu_ragged.py's every-row parity check), numpy: the
    MeViS-like mix N ~ U[8,80], T ~ U[20,200], L ~ U[4,24] with (a) one expression per video and (b) four expressions per video, drawn
    from ONE generator in this order.  Returns {tag: dict(shapes, lens, sample_video, videos, texts, labels)}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d, D = cfg["object_token_dim"], cfg["lang_token_dim"]
    out = {}
    for tag, per_video in (("one_expression_per_video", 1), ("four_expressions_per_video", 4)):
        V = n_samples // per_video
        shapes = [(int(rng.integers(8, 81)), int(rng.integers(20, 201))) for _ in range(V)]
        lens = [int(rng.integers(4, 25)) for _ in range(n_samples)]
        sample_video = [i // per_video for i in range(n_samples)]
        videos = [rng.standard_normal((n, t, d)).astype(np.float32) for n, t in shapes]
        texts = [rng.standard_normal((ln, D)).astype(np.float32) for ln in lens]
        labels = [(rng.uniform(size=shapes[v][0]) < 0.2).astype(np.float32) for v in sample_video]
        out[tag] = {"per_video": per_video, "shapes": shapes, "lens": lens, "sample_video": sample_video, "videos": videos, "texts": texts,
                    "labels": labels}
    return out


def t_out_lengths(T):
    """Frame count after each encoder conv (three stride-2 k=3 p=1 convs, then stride 1)."""
    lens = []
    t = T
    for k, s, p in ((3, 2, 1), (3, 2, 1), (3, 2, 1), (3, 1, 1), (3, 1, 1), (1, 1, 0)):
        t = (t + 2 * p - k) // s + 1
        lens.append(t)
    return lens


def flops_per_sample(cfg, N, T, L):
    """Algorithmic forward FLOPs of one sample (SURVEY §8d formula)."""
    D, H = cfg["lang_token_dim"], NUM_HEADS
    dh = D // H
    W = L + cfg["n_negative"]
    lens = t_out_lengths(T)
    conv = 0
    for (cin, cout, k, _s, _p), tl in zip(encoder_spec(cfg), lens):
        conv += N * 2 * cin * cout * k * tl
    Tp = lens[-1]
    M = N * Tp
    nl = cfg["n_layers"]
    proj = nl * ((4 + 4 + 2) * 2 * M * D * D + 2 * 2 * W * D * D)
    attn = nl * (Tp * H * 4 * N * N * dh + N * H * 4 * Tp * Tp * dh + H * 4 * M * W * dh)
    score = 2 * M * W * D + 2 * N * W * D
    return {"conv": conv, "proj": proj, "attn": attn, "score": score, "total": conv + proj + attn + score}


def shared_flops_per_video(cfg, N, T):
    """Layer 0's inter-object and motion sub-blocks (q/k/v/out projections + attention cores, module/module.py:31-43): with
    the conv encoder, the part of a sample's forward that does not depend on the text - sola_forward_ragged computes it
    once per video (the "conv" entry of flops_per_sample is the rest of that part)."""
    D, H = cfg["lang_token_dim"], NUM_HEADS
    dh = D // H
    Tp = t_out_lengths(T)[-1]
    M = N * Tp
    return 8 * 2 * M * D * D + Tp * H * 4 * N * N * dh + N * H * 4 * Tp * Tp * dh


def attn_bytes_per_sample(cfg, N, T, L, elem=4):
    """Algorithmic attention-core bytes (q,k,v read once + o written once)."""
    D = cfg["lang_token_dim"]
    W = L + cfg["n_negative"]
    M = N * t_out_lengths(T)[-1]
    return cfg["n_layers"] * elem * (4 * M * D + 4 * M * D + 2 * M * D + 2 * W * D)
