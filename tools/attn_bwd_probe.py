"""Attention backward per shape class, fused one-pass kernel (sola_tune attn_bwd_fused 1, default) vs the two-pass kernels (0):
time per call and fraction of the HBM peak on the algorithmic bytes (q, k, v, o, dO read + dQ, dK, dV written), max difference.

    python tools/attn_bwd_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, ops  # noqa: E402

lib = _lib.lib()
D, H = 1024, 8
for tag, B, N, Tp in (("NS/64", 64, 64, 4), ("N80/64", 64, 80, 4), ("C4/16", 16, 128, 16), ("N40 T'12", 64, 40, 12), ("N16 T'25", 64, 16, 25)):
    M = B * N * Tp
    q, k, v, do = (torch.randn(M, D, device="cuda") for _ in range(4))
    line = []
    for name, G, Sq, Sk, inner, qa in ((f"obj {N}x{N}", B * Tp, N, N, Tp, (N * Tp, 1, Tp)), (f"motion {Tp}x{Tp}", B * N, Tp, Tp, 1, (Tp, 0, 1))):
        o, lse = ops.attention(q, k, v, G, H, Sq, Sk, inner, qa, qa, return_lse=True)
        res, t = {}, {}
        for fused in (1, 0, 1, 0, 11, 12):  # 11 / 12: the one-pass kernel without its tile arithmetic / with only the first tile staged
            _lib.check(lib.sola_tune(b"attn_bwd_fused", 1 if fused else 0), "tune")
            _lib.check(lib.sola_tune(b"attn_bwd_ablate", fused - 10 if fused > 1 else 0), "tune")
            res[fused] = ops.attention_backward(q, k, v, o, do, lse, G, H, Sq, Sk, inner, qa, qa)
            torch.cuda.synchronize()
            _lib.profile_enable(True); _lib.profile_read(reset=True)
            for _ in range(10):
                ops.attention_backward(q, k, v, o, do, lse, G, H, Sq, Sk, inner, qa, qa)
            torch.cuda.synchronize()
            ms = _lib.profile_read(reset=True)["attn_bwd"]["ms"] / 10  # kernels only (HIP events around the launches)
            _lib.profile_enable(False)
            t[fused] = min(t.get(fused, 1e9), ms)
        diff = max(float((a - b).abs().max()) for a, b in zip(res[1], res[0]))
        ref = max(float(b.abs().max()) for b in res[0])
        nbytes = 8 * M * D * 4
        line.append(f"{name}: fused {t[1] * 1e3:6.1f} us ({nbytes / t[1] / 1e6 / 8000 * 100:4.1f}%) two-pass {t[0] * 1e3:6.1f} us [no arithmetic {t[11] * 1e3:6.1f}, one staging {t[12] * 1e3:6.1f}]  maxdiff {diff:.1e} of {ref:.1e}")
    print(f"{tag:9s} " + " | ".join(line), flush=True)
_lib.check(lib.sola_tune(b"attn_bwd_fused", 1), "tune")
_lib.check(lib.sola_tune(b"attn_bwd_ablate", 0), "tune")
