#!/usr/bin/env python3
"""The bench's C4 stress leg (T=128, N=128, 32 samples) under sola_tune settings: attention routing of the 128-key units.
usage: stress_probe.py key=val[,key=val] ..."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sola_amd import synth, _lib
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda().eval(); m.precision = "f16x3"; m.ws_policy = "always"
for setting in (sys.argv[1:] or ["attn_split_min_keys=64"]) * 2:
    for kv in setting.split(","):
        k, v = kv.split("=")
        _lib.check(_lib.lib().sola_tune(k.encode(), int(v)), "tune")
    r = bench.stress_leg(cfg, m, torch.device("cuda", 0), "f16x3", 6)
    print(setting, r["value"], r["ms_per_step"], r["roofline_attention"]["frac"], r["kernel_ms_per_step"])
