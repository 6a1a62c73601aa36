"""Differential fuzz of the split-input attention (attn_fwd_spin_kernel / attn.hip's kernel behind sola_attention_split) against the
exact-f32 attention on the same values: random (samples, tracks, steps, text length), inter-object and object -> language layouts,
f32 and split-f16 output.  Exits non-zero above 2e-5.

    python tools/attn_spin_fuzz.py [cases = 80] [seed = 0]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import ops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
D, H = 1024, 8
worst = 0.0
for case in range(n_cases):
    B, N, Tp, Wn = int(rng.integers(1, 4)), int(rng.integers(17, 129)), int(rng.integers(1, 9)), int(rng.integers(17, 65))
    M = B * N * Tp
    q, k, v = (torch.randn(M, D, device="cuda") for _ in range(3))
    lk, lv = torch.randn(B * Wn, D, device="cuda"), torch.randn(B * Wn, D, device="cuda")
    qs, ks, vs, lks, lvs = (ops.cast_sp16(t) for t in (q, k, v, lk, lv))
    for name, f32in, spin, geo in (("obj", (q, k, v), (qs, ks, vs), (B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp))),
                                   ("o2l", (q, lk, lv), (qs, lks, lvs), (B, H, N * Tp, Wn, 1, (N * Tp, 0, 1), (Wn, 0, 1)))):
        ref = ops.attention(*f32in, *geo)
        for out_split in (False, True):
            got = ops.attention_split(*spin, *geo, out_split=out_split)
            got = ops.decode_sp16(got) if out_split else got
            err = float((got - ref).abs().max())
            worst = max(worst, err)
            if err > 2e-5 or not torch.isfinite(got).all():
                print(f"MISMATCH case {case} B={B} N={N} Tp={Tp} Wn={Wn} {name} out_split={out_split}: {err:.2e}")
                sys.exit(1)
print(f"{n_cases} random cases x 2 layouts x 2 output formats: split-input attention == exact-f32 attention within {worst:.1e}")
