"""What bfloat16 mixed precision does to ONE training step of this network at random-init weights, measured on the CPU oracle
(test infrastructure; no GPU, no library): autograd through oracle/sola_oracle.py in float32 against the same step under
torch.autocast("cpu", dtype=torch.bfloat16) - the textbook bf16 training recipe.  Gives the error class the library's "bf16"
mode (bfloat16 GEMM operands, f32 storage) is to be compared with: tests/test_gpu_backward.py LOWP_TRAIN_TOL.

    python tools/bf16_autocast_oracle.py            # the shape of test_f16_operand_training_vs_exact_f32: 8 x 40 x 32 x 10, seed 77
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sola_oracle  # noqa: E402
from sola_amd import synth  # noqa: E402

cfg = synth.DEFAULT_MODEL_CFG
B, N, T, L = 8, 40, 32, 10
sd = synth.make_state_dict(cfg, 42)
inp = synth.make_inputs(cfg, B, N, T, L, 77)


def step(autocast):
    tsd = {k: torch.tensor(v, requires_grad=(k != "positional_encoding_gaussian_matrix")) for k, v in sd.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        sm, st = sola_oracle.forward(tsd, cfg, inp["object_tokens"], inp["lang_tokens"])
    neg = tsd["negative_token.weight"].unsqueeze(0).repeat(B, 1, 1)
    ls = sola_oracle.losses(sm.float(), st.float(), inp["labels"], inp["pos_tokens"], neg, 1.5, 0.07, 0.3)
    ls["total"].backward()
    return float(ls["total"]), sm.detach().float(), {k: v.grad.double() for k, v in tsd.items() if v.grad is not None}


l32, sm32, g32 = step(False)
l16, sm16, g16 = step(True)
total = math.sqrt(sum(float(v.pow(2).sum()) for v in g32.values()))
n16 = math.sqrt(sum(float(v.pow(2).sum()) for v in g16.values()))
cos = sum(float((g16[k] * g32[k]).sum()) for k in g32) / (total * n16)
rel = sorted(float((g16[k] - g32[k]).norm()) / (float(g32[k].norm()) + 1e-5 * total) for k in g32)
print(f"oracle fp32 vs oracle under torch.autocast(bfloat16), B,N,T,L = {B},{N},{T},{L}:")
print(f"  loss {l32:.5f} vs {l16:.5f} (rel {abs(l16 / l32 - 1):.3e}); worst logit difference {float((sm16 - sm32).abs().max()):.3f}")
print(f"  gradient cosine {cos:.5f}; per-tensor relative error worst {rel[-1]:.3e}, median {rel[len(rel) // 2]:.3e}; norm ratio {n16 / total:.4f}")
