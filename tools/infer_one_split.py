import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from sola_amd import _lib, synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().eval(); m.precision = "f32"
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, 1, 64, 32, 16, 1).items()}
def call():
    with torch.no_grad(): return m(inp["object_tokens"], inp["lang_tokens"])
for pol in ("always", "cached"):
    m.ws_policy = pol
    for _ in range(20): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): call()
    t_host = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 200
    _lib.profile_enable(True); _lib.profile_read(True)
    for _ in range(50): call()
    torch.cuda.synchronize(); prof = _lib.profile_read(True); _lib.profile_enable(False)
    k = {c: (round(v["ms"] / 50 * 1e3, 1), v["launches"] // 50) for c, v in prof.items() if v["launches"]}
    print(pol, "wall/call us", round(wall * 1e6, 1), "host enqueue us", round(t_host * 1e6, 1), "kernel us (launches)", k, "sum", round(sum(v[0] for v in k.values()), 1), sum(v[1] for v in k.values()))
