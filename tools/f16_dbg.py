#!/usr/bin/env python3
"""Per-stage relative error of the 16-bit storage mode against the exact-f32 mode (workspace taps), to tell accumulated f16
rounding (smooth growth, ~1e-3 per stage) from a defect (a jump)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import synth
from sola_amd.module import LanguageAlignedTrackSelectionModule

cfg = synth.DEFAULT_MODEL_CFG
B, N, T, L = 1, 64, 32, 16
sd = synth.make_state_dict(cfg, 42)
inp = synth.make_inputs(cfg, B, N, T, L, 201)
obj, lang = torch.from_numpy(inp["object_tokens"]).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda()
taps = {}
for prec in ("f32", "f16", "f16x3"):
    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.cuda().eval(); m.precision = prec
    with torch.no_grad():
        sm, st = m(obj, lang)
    taps[prec] = {"score_map": sm.cpu().numpy(), "score_tokens": st.cpu().numpy()}
    names = [f"conv{i}" for i in range(5)] + ["l0_obj", "l0_motion", "l0_o2l", "l1_obj", "l1_motion"]
    if prec == "f16":
        names += ["conv5_sp"]
    elif prec == "f32":
        names += ["conv5"]
    for nme in names:
        try:
            t = m.workspace_tap(nme)
        except Exception as e:
            continue
        if prec == "f16":
            rows, cols = t.shape
            t = t.reshape(-1).view(torch.float16)[: rows * cols].reshape(rows, cols).float()
        taps[prec][nme.replace("_sp", "")] = t.cpu().numpy()
for k in taps["f32"]:
    ref = taps["f32"][k]
    line = f"{k:12s} |ref| rms {np.sqrt((ref**2).mean()):9.3e} max {np.abs(ref).max():9.3e}"
    for prec in ("f16",):
        if k in taps[prec] and taps[prec][k].shape == ref.shape:
            e = taps[prec][k] - ref
            line += f"   {prec}: rms err {np.sqrt((e**2).mean()):9.3e}  max err {np.abs(e).max():9.3e}  rel rms {np.sqrt((e**2).mean())/np.sqrt((ref**2).mean()):8.2e}"
    print(line)
