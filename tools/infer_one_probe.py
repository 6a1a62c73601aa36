"""One sample per forward call (the reference's inference.py / evaluator batch size): wall time per call by precision mode and batch size.
usage: infer_one_probe.py [key=value sola_tune switches]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
for kv in sys.argv[1:]:
    k, v = kv.split("="); _lib.check(_lib.lib().sola_tune(k.encode(), int(v)), kv)
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().eval()
for (B, N, T, L) in [(1, 64, 32, 16), (1, 44, 110, 12), (4, 64, 32, 16), (8, 64, 32, 16)]:
    inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, N, T, L, 1).items()}
    line = f"B={B} N={N} T={T} L={L}:"
    for prec in ("f16x3", "f32", "f16"):
        m.precision = prec
        with torch.no_grad():
            for _ in range(10): m(inp["object_tokens"], inp["lang_tokens"])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): m(inp["object_tokens"], inp["lang_tokens"])
            torch.cuda.synchronize()
        line += f"  {prec} {(time.perf_counter() - t0) * 10:.3f} ms"
    print(line, flush=True)
