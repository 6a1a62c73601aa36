"""The three attention calls of an alignment layer through the DEFAULT dispatch, per shape class: time and fraction of the
HBM peak on the algorithmic bytes (q, k, v read + o written, 4 B per element).  HIP events, standalone.

    python tools/attn_default_probe.py [tune key=value,...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, ops  # noqa: E402

lib = _lib.lib()
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 else []):
    k, v = kv.split("=")
    _lib.check(lib.sola_tune(k.encode(), int(v)), kv)
D, H, Wn = 1024, 8, 48
for tag, B, N, Tp in (("NS", 256, 64, 4), ("N80", 256, 80, 4), ("C4", 32, 128, 16), ("N16", 512, 16, 4)):
    M = B * N * Tp
    q, k, v = (torch.randn(M, D, device="cuda") for _ in range(3))
    lk, lv = torch.randn(B * Wn, D, device="cuda"), torch.randn(B * Wn, D, device="cuda")
    cases = {
        f"obj Sq=Sk={N}": (lambda: ops.attention(q, k, v, B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp)), 4 * M * D * 4),
        f"motion Sq=Sk={Tp}": (lambda: ops.attention(q, k, v, B * N, H, Tp, Tp, 1, (Tp, 0, 1), (Tp, 0, 1)), 4 * M * D * 4),
        f"o2l Sq={N * Tp} Sk=48": (lambda: ops.attention(q, lk, lv, B, H, N * Tp, Wn, 1, (N * Tp, 0, 1), (Wn, 0, 1)), (2 * M + 2 * B * Wn) * D * 4),
    }
    tot_t, tot_b = 0.0, 0
    line = []
    for name, (fn, nbytes) in cases.items():
        best = 1e9
        for _ in range(3):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        tot_t += best; tot_b += nbytes
        line.append(f"{name}: {best * 1e3:6.1f} us {nbytes / best / 1e6 / 8000 * 100:4.1f}%")
    print(f"{tag:4s} " + " | ".join(line) + f" | all three {tot_b / tot_t / 1e6 / 8000 * 100:4.1f}%", flush=True)
    del q, k, v, lk, lv
