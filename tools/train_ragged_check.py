"""Split-f16 training path on ragged sizes (rows not multiples of the tile sizes): per-parameter gradient agreement with the
exact-f32 path in the Frobenius norm."""
import sys, math, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_backward as tb
from sola_amd import synth, _lib
m, _ = tb.build(synth.DEFAULT_MODEL_CFG)
cfg = synth.DEFAULT_MODEL_CFG
_lib.lib().sola_tune(b"train_split_min_rows", 0)
for (B, N, T, L) in [(7, 37, 32, 10), (3, 5, 17, 3), (1, 130, 40, 9), (2, 64, 128, 16), (11, 9, 9, 1)]:
    gr = {}
    for prec in ("f32", "f16x3"):
        m.precision = prec
        _, loss3, g = tb.train_step_grads(m, cfg, B, N, T, L, 5)
        gr[prec] = {k: v.clone() for k, v in g.items()}
    total = math.sqrt(sum(float(v.double().pow(2).sum()) for v in gr["f32"].values()))
    worst = max(((float((gr["f16x3"][k] - v).double().norm()) / (float(v.double().norm()) + 1e-6 * total), k) for k, v in gr["f32"].items()))
    print((B, N, T, L), "loss", [round(x, 5) for x in loss3.tolist()], "worst relative gradient difference %.2e at %s" % worst, flush=True)
