"""rocprofv3 target: one sample per optimizer step through module.train_step(optimizer=...) (exact f32, N=64 T=32 L=16), 100 steps.
    tools/prof_stats.sh one1 tools/train_one_target.py [tune key=value,...]; python tools/trace_gaps.py gpurun_out/prof_one1/stats_kernel_trace.csv"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 else []):
    k, v = kv.split("=")
    _lib.check(_lib.lib().sola_tune(k.encode(), int(v)), kv)
cfg = synth.DEFAULT_MODEL_CFG
N, T, L = 64, 32, 16
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().train(); m.precision = "f32"
opt = torch.optim.AdamW(m.parameters(), lr=1e-5, fused=True)
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, 1, N, T, L, 1).items()}
for _ in range(int(os.environ.get("ONE_STEPS", "100"))):
    m.train_step(inp["object_tokens"], inp["lang_tokens"], inp["labels"], inp["pos_tokens"], 1.5, 0.07, 0.3, max_grad_norm=1.0, optimizer=opt)
torch.cuda.synchronize()
print("done")
