"""Per-parameter gradient error of the reduced-precision training modes against the exact-f32 path (same inputs, dropout mask
and weights): relative Frobenius error per tensor, cosine of the whole gradient, losses.  -> profiles/r02_f16_training_errors.log"""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import synth
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda().train()
B, N, T, L = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 40, 32, 10)
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, N, T, L, 77).items()}
res = {}
for prec in ("f32", "f16x3", "f16"):
    m.precision = prec
    m.zero_grad(set_to_none=True)
    if hasattr(m, "set_dropout_seed"): m.set_dropout_seed(1234)
    torch.manual_seed(0)
    sm, st = m(inp["object_tokens"], inp["lang_tokens"])
    neg = m.negative_token.weight.clone().unsqueeze(0).repeat(B, 1, 1)
    l3 = track_selection_losses(sm, st, inp["labels"], inp["pos_tokens"], neg, 1.5, 0.07, 0.3)
    l3[0].backward()
    torch.cuda.synchronize()
    res[prec] = (l3.detach().cpu().numpy(), {k: p.grad.detach().double().clone() for k, p in m.named_parameters()}, sm.detach().clone())
ref = res["f32"][1]
tot = math.sqrt(sum(float(v.pow(2).sum()) for v in ref.values()))
print(f"shape B={B} N={N} T={T} L={L}; total |g| {tot:.4e}; losses f32 {res['f32'][0]}")
for prec in ("f16x3", "f16"):
    l3, g, sm = res[prec]
    dot = sum(float((g[k] * ref[k]).sum()) for k in ref)
    n2 = math.sqrt(sum(float(v.pow(2).sum()) for v in g.values()))
    errs = sorted(((float((g[k] - ref[k]).norm()) / (float(ref[k].norm()) + 1e-6 * tot), k) for k in ref), reverse=True)
    whole = math.sqrt(sum(float((g[k] - ref[k]).pow(2).sum()) for k in ref)) / tot
    print(f"== {prec}: losses {l3}  max|dlogit| {float((sm - res['f32'][2]).abs().max()):.3e}  cos(g, g_f32) {dot / (n2 * tot):.6f}  |g - g_f32| / |g_f32| {whole:.3e}")
    for e, k in errs[:12]: print(f"   {e:9.3e}  {k}")
    print(f"   median {errs[len(errs) // 2][0]:.3e}")
