"""Ragged training step (forward_ragged(differentiable) + per-sample losses + backward + clip + AdamW) on a MeViS-like mix of
shapes (N~U[8,80], T~U[20,200], L~U[4,24], seed 2024), per precision mode, with the in-library per-kernel breakdown.

    python tools/train_ragged_probe.py [samples per step = 64] [modes, comma separated = f32,f16x3,f16] [tune key=value,...]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, synth  # noqa: E402
from sola_amd.loss import track_selection_losses_ragged  # noqa: E402
from sola_amd.module import LanguageAlignedTrackSelectionModule, collate_ragged  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f32", "f16x3", "f16"]
for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 else []):  # sola_tune switches for A/B runs: key=value,...
    k, v = kv.split("=")
    _lib.check(_lib.lib().sola_tune(k.encode(), int(v)), kv)
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-5, fused=True)
samples = synth.make_ragged_samples(cfg, S, 2024, "cuda")
objs, langs = collate_ragged([s["obj"] for s in samples]), collate_ragged([s["lang"] for s in samples])  # views of one buffer, as a collate function hands them over
labels = torch.cat([s["labels"] for s in samples])
pos = torch.stack([s["pos"] for s in samples])
rows = sum(int(o.shape[0] * o.shape[1]) for o in objs)
flops = sum(synth.flops_per_sample(cfg, int(o.shape[0]), int(o.shape[1]), int(l.shape[0]))["total"] for o, l in zip(objs, langs))
print(f"{S} samples, {rows} object-token rows, {3 * flops / S / 1e9:.1f} GFLOP per sample (forward + backward)", flush=True)


def step():
    opt.zero_grad(set_to_none=True)
    m.forward_ragged(objs, langs)
    flat, tok, offs, counts = m.last_ragged
    loss = track_selection_losses_ragged(flat, tok, labels, pos, m.negative_token.weight, offs, counts, 1.5, 0.07, 0.3)
    loss[:, 0].mean().backward()
    m.clip_grad_norm_(1.0)
    opt.step()
    return loss


for prec in modes:
    m.precision = prec
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        ls = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    _lib.profile_enable(True)
    _lib.profile_read(reset=True)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_read(reset=True)
    _lib.profile_enable(False)
    kms = {k: (round(v["ms"] / 4, 3), v["launches"] // 4) for k, v in prof.items() if v["launches"]}
    print(f"precision {prec}: {dt * 1e3:.2f} ms/step  {S / dt:.0f} samples/s  {3 * flops / dt / 1e12:.0f} model TFLOP/s  mean loss {float(ls[:, 0].mean()):.4f}", flush=True)
    print("   kernel ms per step (launches):", kms, flush=True)
