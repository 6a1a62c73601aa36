"""rocprofv3 target: one sample per inference call (the reference's inference.py:58 / evaluator batch size), N=64 T=32 L=16, default mode.
    tools/prof_stats.sh inf1 tools/infer_one_target.py [precision = f16x3] [N] [T]; python tools/trace_gaps.py gpurun_out/prof_inf1/stats_kernel_trace.csv 50 score_head_kernel"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().eval(); m.precision = prec
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, 1, N, T, 16, 1).items()}
with torch.no_grad():
    for _ in range(200):
        m(inp["object_tokens"], inp["lang_tokens"])
torch.cuda.synchronize()
print("done")
