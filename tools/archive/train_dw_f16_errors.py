"""Gradient error of the split-f16 training step against the exact-f32 step, with the weight-gradient products on split-f16 pairs
(sola_tune train_dw_f16 0) and on plain f16 operands (1): per-tensor relative Frobenius error (worst / median), cosine of the whole
gradient, and the tensors that move most.  Shape of tests/test_gpu_backward.py::test_split_training_gradients_match_exact_f32 and a
larger batch."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, synth
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda().eval()
def grads(B, N, T, L, seed):
    c = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, N, T, L, seed).items()}
    m.zero_grad(set_to_none=True)
    sm, st = m(c["object_tokens"], c["lang_tokens"])
    neg = m.negative_token.weight.clone().unsqueeze(0).repeat(B, 1, 1)
    track_selection_losses(sm, st, c["labels"], c["pos_tokens"], neg, 1.5, 0.07, 0.3)[0].backward()
    torch.cuda.synchronize()
    return {k: p.grad.double().clone() for k, p in m.named_parameters()}
for shape in ((8, 40, 32, 10, 77), (64, 64, 32, 16, 5)):
    m.precision = "f32"
    ref = grads(*shape)
    total = math.sqrt(sum(float(v.pow(2).sum()) for v in ref.values()))
    for dw in (0, 1):
        _lib.check(_lib.lib().sola_tune(b"train_dw_f16", dw), "tune")
        m.precision = "f16x3"
        g = grads(*shape)
        rel = sorted(((float((g[k] - ref[k]).norm()) / (float(ref[k].norm()) + 1e-6 * total), k) for k in ref), reverse=True)
        n = math.sqrt(sum(float(v.pow(2).sum()) for v in g.values()))
        cos = sum(float((g[k] * ref[k]).sum()) for k in ref) / (total * n)
        wrel = [r for r, k in rel if k.endswith("weight") and ref[k].dim() >= 2]
        print(f"B,N,T,L={shape[:4]} train_dw_f16={dw}: cosine {cos:.7f}; per-tensor rel. Frobenius error worst {rel[0][0]:.2e} ({rel[0][1]}), "
              f"median {rel[len(rel) // 2][0]:.2e}; weight matrices worst {max(wrel):.2e} median {sorted(wrel)[len(wrel) // 2]:.2e}", flush=True)
_lib.check(_lib.lib().sola_tune(b"train_dw_f16", 0), "tune")
