"""Training step time (forward_train + loss + backward + clip + AdamW) at a batch, per precision mode.

    python tools/train_probe.py [batch = 64] [modes, comma separated = f32,f16x3,f16,bf16] [sola_tune settings: key=value,...]
"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import synth
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if len(sys.argv) > 3:
    from sola_amd import _lib
    for kv in sys.argv[3].split(","):
        k, v = kv.split("=")
        _lib.check(_lib.lib().sola_tune(k.encode(), int(v)), "tune " + k)
        print("tune", k, v)
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-5, fused=True)
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, 64, 32, 16, 1).items()}
def step():
    opt.zero_grad(set_to_none=True)
    sm, st = m(inp["object_tokens"], inp["lang_tokens"])
    neg = m.negative_token.weight.clone().unsqueeze(0).repeat(B, 1, 1)
    loss3 = track_selection_losses(sm, st, inp["labels"], inp["pos_tokens"], neg, 1.5, 0.07, 0.3)
    loss3[0].backward()
    m.clip_grad_norm_(1.0)
    opt.step()
    return loss3
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f32", "f16x3", "f16", "bf16"]
for prec in modes + modes:
    m.precision = prec
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"precision {prec}: {dt*1e3:.2f} ms/step  {B/dt:.0f} samples/s  loss {float(l[0]):.5f}", flush=True)
