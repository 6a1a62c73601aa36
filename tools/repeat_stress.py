#!/usr/bin/env python3
"""Run-to-run repeatability of the whole path (no kernel uses float atomics, so every output must repeat bit for bit): the
headline forward in the three precision modes, a ragged forward, and a 64-sample training step's gradients."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sola_amd import synth
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = synth.DEFAULT_MODEL_CFG
sd = synth.make_state_dict(cfg, 42)
bad = 0


def bits(t):
    return t.contiguous().view(torch.int32)


m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
m = m.cuda().eval()
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, 256, 64, 32, 16, 31).items()}
for prec in ("f32", "f16x3", "f16"):
    m.precision = prec
    first, differ = None, 0
    for _ in range(reps):
        with torch.no_grad():
            sm, st = m(inp["object_tokens"], inp["lang_tokens"])
        cur = (bits(sm).clone(), bits(st).clone())
        if first is None:
            first = cur
        elif not (torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])):
            differ += 1
    print(f"forward B=256 {prec}: {differ} of {reps - 1} repeats differ")
    bad += differ

# ragged forward (MeViS-like mix)
import numpy as np
rng = np.random.default_rng(5)
videos = [torch.from_numpy(rng.standard_normal((int(rng.integers(8, 81)), int(rng.integers(20, 201)), 256), dtype=np.float32)).cuda() for _ in range(24)]
texts = [torch.from_numpy(rng.standard_normal((int(rng.integers(4, 25)), 1024), dtype=np.float32)).cuda() for _ in range(48)]
sample_video = [i % 24 for i in range(48)]
for prec in ("f32", "f16x3"):
    m.precision = prec
    first, differ = None, 0
    for _ in range(reps):
        with torch.no_grad():
            m.forward_ragged(videos, texts, sample_video)
        flat, tok = m.last_ragged[0], m.last_ragged[1]
        cur = (bits(flat).clone(), bits(tok).clone())
        if first is None:
            first = cur
        elif not (torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])):
            differ += 1
    print(f"ragged forward, 48 samples {prec}: {differ} of {reps - 1} repeats differ")
    bad += differ

# training step gradients
B = 64
tinp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, 64, 32, 16, 1).items()}
for prec in ("f32", "f16x3", "f16", "bf16"):
    mt = LanguageAlignedTrackSelectionModule(cfg)
    mt.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    mt = mt.cuda().train(); mt.precision = prec
    first, differ = None, 0
    for it in range(max(5, reps // 2 + 1)):
        for p in mt.parameters():
            p.grad = None
        torch.manual_seed(11)
        sm, st = mt(tinp["object_tokens"], tinp["lang_tokens"])
        neg = mt.negative_token.weight.clone().unsqueeze(0).repeat(B, 1, 1)
        loss3 = track_selection_losses(sm, st, tinp["labels"], tinp["pos_tokens"], neg, 1.5, 0.07, 0.3)
        loss3[0].backward()
        cur = [bits(p.grad).clone() for p in mt.parameters() if p.grad is not None]
        if it == 0:
            continue  # the first step of a 16-bit mode sizes the kept-operand arena: its casts (bf16: its storage) take another route
        if first is None:
            first = cur
        elif not all(torch.equal(a, b) for a, b in zip(cur, first)):
            differ += 1
    print(f"training step B=64 {prec}: gradients differ in {differ} of {max(5, reps // 2 + 1) - 2} repeats")
    bad += differ
    del mt

# ragged training step (the bf16 step's storage route, its bf16-MFMA attention kernels, the staggered GEMM starts)
from sola_amd.loss import track_selection_losses_ragged
samples = synth.make_ragged_samples(cfg, 32, 2024, "cuda")
objs, langs = [x["obj"] for x in samples], [x["lang"] for x in samples]
labels = torch.cat([x["labels"] for x in samples]); pos = torch.stack([x["pos"] for x in samples])
for prec in ("bf16", "f16x3"):
    mt = LanguageAlignedTrackSelectionModule(cfg)
    mt.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    mt = mt.cuda().train(); mt.precision = prec
    first, differ = None, 0
    n = max(5, reps // 2)
    for it in range(n):
        for p in mt.parameters():
            p.grad = None
        torch.manual_seed(11)
        mt.forward_ragged(objs, langs, differentiable=True)
        flat, tok, offs, counts = mt.last_ragged
        loss = track_selection_losses_ragged(flat, tok, labels, pos, mt.negative_token.weight, offs, counts, 1.5, 0.07, 0.3)
        loss[:, 0].mean().backward()
        cur = [bits(p.grad).clone() for p in mt.parameters() if p.grad is not None]
        if it == 0:
            continue  # the first step sizes the kept-operand arena: its casts take another route
        if first is None:
            first = cur
        elif not all(torch.equal(a, b) for a, b in zip(cur, first)):
            differ += 1
    print(f"ragged training step, 32 samples {prec}: gradients differ in {differ} of {n - 2} repeats")
    bad += differ
    del mt
sys.exit(1 if bad else 0)
