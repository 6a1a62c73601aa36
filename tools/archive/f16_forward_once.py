"""The 16-bit storage mode's forward alone (NS shape, 256 samples; or B N T from argv), for rocprofv3: tools/prof_stats.sh f16 tools/f16_forward_once.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sola_amd import synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
B, N, T = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 64, 32)
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda().eval(); m.precision = "f16"; m.ws_policy = "always"
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, N, T, 16, 1).items()}
with torch.no_grad():
    for _ in range(12):
        m(inp["object_tokens"], inp["lang_tokens"])
torch.cuda.synchronize()
print("fallbacks", m.split_fallbacks())
