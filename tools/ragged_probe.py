#!/usr/bin/env python3
"""The bench's ragged legs alone (MeViS-like mix, 128 samples per launch), with the per-kernel split."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sola_amd import synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda().eval(); m.precision = sys.argv[1] if len(sys.argv) > 1 else "f16x3"; m.ws_policy = "always"
r = bench.ragged_leg(cfg, m, torch.device("cuda", 0), 6, 288.0)
for k in ("one_expression_per_video", "four_expressions_per_video"):
    print(k, json.dumps({kk: r[k][kk] for kk in ("value", "ms_per_launch", "model_tflops_executed", "executed_vs_uniform_batch_model_tflops", "kernel_ms_per_launch")}))
