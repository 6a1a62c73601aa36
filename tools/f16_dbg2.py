import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
so = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
sd = synth.make_state_dict(cfg, 42)
inp = synth.make_inputs(cfg, 1, 64, 32, 16, 300)
obj, lang = torch.from_numpy(inp["object_tokens"] * np.float32(so)).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda()
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
m = m.cuda().eval(); m.precision = "f16"; m.split_guard = False
with torch.no_grad():
    sm, st = m(obj, lang)
print("fallbacks", m.split_fallbacks(), "finite", bool(torch.isfinite(sm).all()))
for nme in ["obj_sp", "conv0", "act0", "conv1", "act1", "conv4", "act4", "conv5_sp", "q", "k", "v", "attn", "res", "l0_obj", "l0_motion", "l0_o2l", "l1_motion", "lang_sp", "lk"]:
    try:
        t = m.workspace_tap(nme)
    except Exception as e:
        print(nme, "n/a"); continue
    rows, cols = t.shape
    h = t.reshape(-1).view(torch.float16)[: rows * cols].float()
    print(f"{nme:10s} absmax {h.abs().max().item():10.3e} rms {h.pow(2).mean().sqrt().item():10.3e} nonfinite {int((~torch.isfinite(h)).sum())}")
