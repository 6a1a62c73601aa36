"""One sample per optimizer step (the reference's configs/mevis/default.yaml batch_size 1): wall time per step, the kernels' own time
(in-library HIP events) and the host-side pieces, exact f32 (below 1024 token rows every mode runs these kernels)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, synth
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule
for kv in sys.argv[1:]:  # sola_tune switches for A/B runs: key=value
    k, v = kv.split("="); _lib.check(_lib.lib().sola_tune(k.encode(), int(v)), kv)
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().train(); m.precision = "f32"
opt = torch.optim.AdamW(m.parameters(), lr=1e-5, fused=True)
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, 1, 64, 32, 16, 1).items()}
acc = {}
def tick(name, t0, sync=False):
    if sync: torch.cuda.synchronize()
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + t - t0; return t
def step(timed):
    t = time.perf_counter()
    opt.zero_grad(set_to_none=True); t = tick("zero_grad", t) if timed else t
    sm, st = m(inp["object_tokens"], inp["lang_tokens"]); t = tick("forward (host)", t) if timed else t
    neg = m.negative_token.weight.clone().unsqueeze(0)
    loss3 = track_selection_losses(sm, st, inp["labels"], inp["pos_tokens"], neg, 1.5, 0.07, 0.3); t = tick("loss (host)", t) if timed else t
    loss3[0].backward(); t = tick("backward (host)", t) if timed else t
    m.clip_grad_norm_(1.0); t = tick("clip (host)", t) if timed else t
    opt.step(); t = tick("adamw (host)", t) if timed else t
for _ in range(10): step(False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step(False)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 50
for _ in range(50): step(True)
torch.cuda.synchronize()
_lib.profile_enable(True); _lib.profile_read(reset=True)
for _ in range(20): step(False)
torch.cuda.synchronize()
prof = _lib.profile_read(reset=True); _lib.profile_enable(False)
kms = sum(v["ms"] for v in prof.values()) / 20
print(f"wall {wall * 1e3:.3f} ms/step; library kernels {kms:.3f} ms/step in {sum(v['launches'] for v in prof.values()) // 20} launches; "
      f"host enqueue time per step: " + ", ".join(f"{k} {v / 50 * 1e3:.3f}" for k, v in acc.items()) + f" = {sum(acc.values()) / 50 * 1e3:.3f} ms")
