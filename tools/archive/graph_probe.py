"""Single-sample latency: eager calls vs a torch.cuda.CUDAGraph (hipGraph) replay of module forward + losses + select."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from sola_amd import synth, ops
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda().eval()
from sola_amd import _lib
if len(sys.argv) > 1: _lib.lib().sola_tune(b"gemm_splitk", int(sys.argv[1]))
for prec in ("f16x3", "f32"):
    m.precision = prec
    for (B, N, T) in [(1, 64, 32), (1, 16, 32), (1, 128, 128), (4, 64, 32), (8, 64, 32), (16, 64, 32)]:
        inp = synth.make_inputs(cfg, B, N, T, 16, 3)
        obj, lang = torch.from_numpy(inp["object_tokens"]).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda()
        labels, pos = torch.from_numpy(inp["labels"]).cuda(), torch.from_numpy(inp["pos_tokens"]).cuda()
        def step():
            with torch.no_grad():
                sm, st = m(obj, lang)
                l3 = track_selection_losses(sm, st, labels, pos, m.negative_token.weight, 1.5, 0.07, 0.3)
                return sm, l3, ops.select(sm, 0.5)[1]
        for _ in range(3): ref = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): step()
        torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 50
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            step(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = step()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        ok = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2])
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 50
        print(f"{prec} B={B} N={N} T={T}: eager {eager*1e3:.3f} ms  graph replay {graph*1e3:.3f} ms  identical={ok}")
