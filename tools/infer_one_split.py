"""One sample per inference call (inference.py:58; N=64, T=32, L=16, exact f32): wall per call, host enqueue time, in-library kernel time by
category; and one sample per optimizer step (module.train_step + fused clip/AdamW).  Optional sola_tune switches: key=value ...
    python tools/infer_one_split.py [gemm_small_pre=0]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
for kv in sys.argv[1:]:
    k, v = kv.split("="); _lib.check(_lib.lib().sola_tune(k.encode(), int(v)), kv)
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().eval(); m.precision = "f32"
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, 1, 64, 32, 16, 1).items()}
def call():
    with torch.no_grad(): return m(inp["object_tokens"], inp["lang_tokens"])
for _ in range(20): call()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): call()
t_host = (time.perf_counter() - t0) / 300
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 300
_lib.profile_enable(True); _lib.profile_read(True)
for _ in range(50): call()
torch.cuda.synchronize(); prof = _lib.profile_read(True); _lib.profile_enable(False)
k = {c: (round(v["ms"] / 50 * 1e3, 1), v["launches"] // 50) for c, v in prof.items() if v["launches"]}
print("inference: wall/call us", round(wall * 1e6, 1), "host enqueue us", round(t_host * 1e6, 1), "kernel us (launches)", k, "sum", round(sum(v[0] for v in k.values()), 1), sum(v[1] for v in k.values()))
m.train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-5, fused=True)
def step():
    m.train_step(inp["object_tokens"], inp["lang_tokens"], inp["labels"], inp["pos_tokens"], 1.5, 0.07, 0.3, max_grad_norm=1.0, optimizer=opt, write_back_grads=False)
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print("one-sample optimizer step ms", round((time.perf_counter() - t0) * 5, 4))
