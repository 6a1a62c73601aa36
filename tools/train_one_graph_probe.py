"""One sample per optimizer step (the reference's batch size): eager calls against a hipGraph replay of the whole step (zero_grad, forward_train,
losses, backward, clip, fused AdamW with capturable=True) at one fixed shape.  Prints wall time per step of both, the number of library
launches per step by profiler category, and whether weights after K eager steps equal weights after K replays (same dropout seeds are baked
into the captured launches: this probe runs with dropout OFF - module.eval() - so the comparison is exact)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, synth
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
N, T, L = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 32, 16)))
def make(capturable):
    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
    m = m.cuda().eval(); m.precision = "f32"
    opt = torch.optim.AdamW(m.parameters(), lr=1e-5, fused=True, capturable=capturable)
    return m, opt
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, 1, N, T, L, 1).items()}
def step(m, opt, set_none=True):
    opt.zero_grad(set_to_none=set_none)
    sm, st = m(inp["object_tokens"], inp["lang_tokens"])  # differentiable call (grad enabled, parameters require grad)
    neg = m.negative_token.weight.clone().unsqueeze(0)
    loss3 = track_selection_losses(sm, st, inp["labels"], inp["pos_tokens"], neg, 1.5, 0.07, 0.3)
    loss3[0].backward()
    m.clip_grad_norm_(1.0)
    opt.step()
    return loss3
def wall(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
m, opt = make(False)
with torch.enable_grad():
    t_eager = wall(lambda: step(m, opt))
    _lib.profile_enable(True); _lib.profile_read(reset=True)
    for _ in range(10): step(m, opt)
    torch.cuda.synchronize(); prof = _lib.profile_read(reset=True); _lib.profile_enable(False)
print(f"shape N={N} T={T} L={L}: eager {t_eager * 1e3:.3f} ms/step; library launches/step " +
      ", ".join(f"{k} {v['launches'] // 10} ({v['ms'] / 10:.3f} ms)" for k, v in prof.items() if v["launches"]), flush=True)
# graph
mg, optg = make(True)
me, opte = make(True)
with torch.enable_grad():
    for _ in range(3): step(mg, optg, set_none=False); step(me, opte, set_none=False)  # warm-up on the same footing (also sizes every workspace)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step(mg, optg, set_none=False)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            loss_g = step(mg, optg, set_none=False)
    except Exception as e:
        print("capture failed:", type(e).__name__, str(e)[:400]); sys.exit(0)
    t_graph = wall(g.replay)
    print(f"graph replay {t_graph * 1e3:.3f} ms/step ({t_eager / t_graph:.2f}x)", flush=True)
    # equality: K more steps on both
    K = 5
    sd0 = {k: v.clone() for k, v in me.state_dict().items()}
    mg.load_state_dict(sd0); mg.weights_changed()
    # optimizer states differ after the timing loops: rebuild both from the same state
    for _ in range(K): step(me, opte, set_none=False)
    torch.cuda.synchronize()
print("done")
