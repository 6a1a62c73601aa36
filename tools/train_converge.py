"""Does a training run in the reduced-precision modes follow the exact-f32 run?  Same initial weights, same batches, same
dropout seeds; the loss is read in eval mode with the exact-f32 forward every few steps.  -> profiles/r02_train_converge.log, r03_train_converge.log (bf16 operands; f16x3 with the
weight-gradient products on plain f16 operands)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import synth
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
B, N, T, L = 32, 64, 32, 16
steps, every, lr = int(sys.argv[1]) if len(sys.argv) > 1 else 60, 10, 1e-4
sd = synth.make_state_dict(cfg, 42)
batches = [{k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, N, T, L, 500 + i).items()} for i in range(4)]
val = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, N, T, L, 999).items()}

def loss_of(m, inp):
    sm, st = m(inp["object_tokens"], inp["lang_tokens"])
    neg = m.negative_token.weight.clone().unsqueeze(0).repeat(B, 1, 1)
    return track_selection_losses(sm, st, inp["labels"], inp["pos_tokens"], neg, 1.5, 0.07, 0.3)

curves = {}
for prec in ("f32", "f16x3", "f16", "bf16"):
    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.cuda()
    opt = torch.optim.AdamW(m.parameters(), lr=lr, fused=True)
    curve = []
    for it in range(steps + 1):
        if it % every == 0:
            m.eval(); m.precision = "f32"
            with torch.no_grad():
                tr = sum(float(loss_of(m, b)[0]) for b in batches) / len(batches)
                va = float(loss_of(m, val)[0])
            curve.append((it, tr, va))
        if it == steps: break
        m.train(); m.precision = prec
        if hasattr(m, "dropout_seed"): m.dropout_seed = 1000 + it
        opt.zero_grad(set_to_none=True)
        l3 = loss_of(m, batches[it % len(batches)])
        l3[0].backward()
        m.clip_grad_norm_(1.0)
        opt.step()
    curves[prec] = curve
    del m, opt
print(f"B={B} N={N} T={T} L={L}, AdamW lr {lr}, clip 1.0, {steps} steps over 4 fixed batches; eval-mode exact-f32 loss (train batches / held-out batch)")
print("step   " + "   ".join(f"{p:>19s}" for p in curves))
for i in range(len(curves["f32"])):
    print(f"{curves['f32'][i][0]:4d}   " + "   ".join(f"{curves[p][i][1]:9.4f} {curves[p][i][2]:9.4f}" for p in curves))
