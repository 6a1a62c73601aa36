"""Round 5: one sample per optimizer step (the reference's batch size) - the call-by-call path through autograd against module.train_step
(sola_train_step: the whole step's launches enqueued from C++): wall time per step, host time per step (enqueue only), per-category kernel time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, synth
from sola_amd.loss import track_selection_losses
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
def step_autograd():
    opt.zero_grad(set_to_none=True)
    sm, st = m(inp["object_tokens"], inp["lang_tokens"])
    neg = m.negative_token.weight.clone().unsqueeze(0)
    loss3 = track_selection_losses(sm, st, inp["labels"], inp["pos_tokens"], neg, 1.5, 0.07, 0.3)
    loss3[0].backward()
    m.clip_grad_norm_(1.0)
    opt.step()
def step_call():
    m.train_step(inp["object_tokens"], inp["lang_tokens"], inp["labels"], inp["pos_tokens"], 1.5, 0.07, 0.3, max_grad_norm=1.0)
    opt.step()
def step_call_adam():
    m.train_step(inp["object_tokens"], inp["lang_tokens"], inp["labels"], inp["pos_tokens"], 1.5, 0.07, 0.3, max_grad_norm=1.0, optimizer=opt)
def step_call_adam_nowb():  # as train.py calls it: an active clip does not rewrite .grad
    m.train_step(inp["object_tokens"], inp["lang_tokens"], inp["labels"], inp["pos_tokens"], 1.5, 0.07, 0.3, max_grad_norm=1.0, optimizer=opt, write_back_grads=False)
def wall(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t2 - t0) / n, (t1 - t0) / n
if os.environ.get("PROBE_SIDE_STREAM") == "1":  # a created (non-null) stream instead of the legacy default stream
    _side = torch.cuda.Stream()
    torch.cuda.set_stream(_side)
    print("running on a created stream")
for name, fn in (("autograd path", step_autograd), ("train_step call", step_call), ("train_step + in-library clip+AdamW", step_call_adam), ("... without gradient write-back", step_call_adam_nowb), ("autograd path", step_autograd), ("train_step call", step_call), ("train_step + in-library clip+AdamW", step_call_adam), ("... without gradient write-back", step_call_adam_nowb)):
    w, h = wall(fn)
    print(f"{name}: wall {w * 1e3:.3f} ms/step ({1 / w:.0f} samples/s), host enqueue {h * 1e3:.3f} ms/step", flush=True)
_lib.profile_enable(True); _lib.profile_read(reset=True)
for _ in range(10): step_call()
torch.cuda.synchronize(); prof = _lib.profile_read(reset=True); _lib.profile_enable(False)
print("library launches/step: " + ", ".join(f"{k} {v['launches'] // 10} ({v['ms'] / 10:.3f} ms)" for k, v in prof.items() if v["launches"]))
