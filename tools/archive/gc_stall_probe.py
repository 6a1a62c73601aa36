"""Why bench.py parks the interpreter's garbage collector: host time of consecutive `forward + loss + select` calls after a
warm-up, with the collector left on vs. parked.  With it on, one call (which one depends on the allocation history, the
first after `_lib.profile_read` in bench.py's sequence) takes ~37 ms on the host - a full collection over torch's object
graph - and the GPU queue runs dry behind it.
    python tools/gc_stall_probe.py [nogc]"""
import gc, sys, time, torch
sys.path.insert(0, "/root/repo")
from sola_amd import _lib, synth, ops
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda().eval(); m.ws_policy = "always"; m.precision = "f16x3"
inp = synth.make_inputs(cfg, 256, 64, 32, 16, seed=1000)
obj, lang = torch.from_numpy(inp["object_tokens"]).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda()
labels, pos = torch.from_numpy(inp["labels"]).cuda(), torch.from_numpy(inp["pos_tokens"]).cuda()
host_ms = []
def step():
    t0 = time.perf_counter()
    with torch.no_grad():
        sm, st = m(obj, lang)
        l3 = track_selection_losses(sm, st, labels, pos, m.negative_token.weight, 1.5, 0.07, 0.3)
        pred = ops.select(sm, 0.5)[1]
    host_ms.append(round((time.perf_counter() - t0) * 1e3, 2))
    return l3, pred
if len(sys.argv) > 1:
    gc.collect(); gc.disable()
for _ in range(3): step()
torch.cuda.synchronize()
_lib.profile_enable(True); _lib.profile_read(reset=True)
t0 = time.perf_counter()
for _ in range(10): keep = step()
torch.cuda.synchronize(); el = (time.perf_counter() - t0) * 1e3
print("gc", "parked" if len(sys.argv) > 1 else "on", "- host ms per call:", host_ms[3:], f"- {el / 10:.2f} ms/step")
