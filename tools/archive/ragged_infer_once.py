"""bench.py's ragged inference leg alone (default dispatch), for rocprofv3: tools/prof_stats.sh ragged tools/ragged_infer_once.py"""
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
    print(k, json.dumps({kk: r[k][kk] for kk in ("value", "ms_per_launch", "kernel_ms_per_launch")}))
