#!/usr/bin/env python3
"""Headline batch, split-f16 mode: attention time per step with the exact-f32 MFMA shape vs the split-f16 MFMA shape (q/k/v
written as split pairs by the projection GEMM) for units of <= 64 keys."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda().eval(); m.precision = "f16x3"; m.ws_policy = "always"
B, N, T, L = 256, 64, 32, 16
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, N, T, L, 1000).items()}
ref = None
for mk, var in ((64, 0), (64, 1), (64, 0), (64, 1)):  # var: attn_splitm off / on
    _lib.check(_lib.lib().sola_tune(b"attn_split_min_keys", mk), "tune")
    _lib.check(_lib.lib().sola_tune(b"attn_splitm", var), "tune")
    with torch.no_grad():
        for _ in range(3): sm, _ = m(inp["object_tokens"], inp["lang_tokens"])
        torch.cuda.synchronize()
        _lib.profile_enable(True); _lib.profile_read(True)
        t0 = time.perf_counter()
        for _ in range(10): sm, _ = m(inp["object_tokens"], inp["lang_tokens"])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    p = _lib.profile_read(True); _lib.profile_enable(False)
    if ref is None: ref = sm.clone()
    print(json.dumps({"attn_split_min_keys": mk, "attn_splitm": var, "ms_per_forward": round(dt * 1e3, 3), "attn_ms": round(p["attn"]["ms"] / 10, 3), "attn_launches": p["attn"]["launches"] // 10,
                      "gemm_ms": round(p["gemm_split256"]["ms"] / 10, 3), "max_diff_vs_first": float((sm - ref).abs().max())}))
