#!/usr/bin/env python3
"""Throughput of the three inference precisions on the uniform batch (NS: 256 x (T=32,N=64); C4: 32 x (T=128,N=128)),
forward + loss + selection, with the attention / GroupNorm / GEMM split of each."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, ops, synth
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule

cfg = synth.DEFAULT_MODEL_CFG
sd = synth.make_state_dict(cfg, 42)
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
m = m.cuda().eval(); m.ws_policy = "always"
for tag, (B, N, T, L) in (("NS", (256, 64, 32, 16)), ("C4", (32, 128, 128, 16))):
    inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, N, T, L, 1000).items()}
    ref = None
    for prec in ("f32", "f16x3", "f16"):
        m.precision = prec
        def step():
            with torch.no_grad():
                sm, st = m(inp["object_tokens"], inp["lang_tokens"])
                track_selection_losses(sm, st, inp["labels"], inp["pos_tokens"], m.negative_token.weight, 1.5, 0.07, 0.3)
                ops.select(sm, 0.5)
            return sm
        for _ in range(3): sm = step()
        torch.cuda.synchronize()
        _lib.profile_enable(True); _lib.profile_read(True)
        t0 = time.perf_counter()
        for _ in range(8): sm = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        prof = _lib.profile_read(True); _lib.profile_enable(False)
        if ref is None: ref = sm.clone()
        a = prof["attn"]
        print(json.dumps({"shape": tag, "precision": prec, "samples_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                          "max_logit_diff_vs_f32": float((sm - ref).abs().max()), "fallbacks": m.split_fallbacks()[0],
                          "attn_GBps": round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1) if a["ms"] else None,
                          "kernel_ms": {k: round(v["ms"] / 8, 3) for k, v in prof.items() if v["launches"]}}))
