"""Profiling target: the inference forward in the 16-bit storage mode (module.precision = "f16") at one uniform shape.

    tools/prof_stats.sh f16c4 tools/f16_target.py [B = 32] [N = 128] [T = 128] [steps = 20] [precision = f16]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import synth  # noqa: E402
from sola_amd.module import LanguageAlignedTrackSelectionModule  # noqa: E402

B, N, T, steps = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 32), (2, 128), (3, 128), (4, 20)))
prec = sys.argv[5] if len(sys.argv) > 5 else "f16"
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().eval()
m.precision = prec
c = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, N, T, 16, seed=1000).items()}
with torch.no_grad():
    for _ in range(steps):
        m(c["object_tokens"], c["lang_tokens"])
torch.cuda.synchronize()
print("done", B, N, T, steps, prec)
