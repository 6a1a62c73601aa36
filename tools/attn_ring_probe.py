"""attn_ring.hip against attn_simple.hip (sola_tune "attn_ring" 1 / 0): bit equality on uniform shapes incl. ragged tails, then the time
of the three attention sites with the ring on / off / also in place of the resident-K/V shape (2).

    python tools/attn_ring_probe.py [check|time|all] [shape tag: NS|N80|C4|N16] [modes, e.g. 01]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, ops  # noqa: E402

lib = _lib.lib()
what = sys.argv[1] if len(sys.argv) > 1 else "all"
D, H = 1024, 8


def tune(v):
    _lib.check(lib.sola_tune(b"attn_ring", v), "tune")


if what in ("check", "all"):
    torch.manual_seed(0)
    for (B, N, Tp) in ((4, 64, 4), (3, 80, 4), (2, 128, 16), (5, 20, 4), (3, 17, 1), (2, 100, 3), (700, 33, 2)):
        M = B * N * Tp
        q, k, v = (torch.randn(M, D, device="cuda") * 2 for _ in range(3))
        outs = {}
        for mode in (1, 0):
            tune(mode)
            outs[mode] = ops.attention(q, k, v, B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp))
        torch.cuda.synchronize()
        same = torch.equal(outs[1], outs[0])
        print(f"obj B={B} N={N} Tp={Tp}: bit-identical {same}  max|diff| {float((outs[1] - outs[0]).abs().max()):.3e}  nan {bool(torch.isnan(outs[1]).any())}", flush=True)
    # many queries against few keys (object -> language): ring mode 2 vs the resident-K/V shape - same arithmetic?  report the difference
    for (B, NT, Wn) in ((6, 256, 48), (3, 320, 21), (2, 1000, 64)):
        q = torch.randn(B * NT, D, device="cuda")
        lk, lv = torch.randn(B * Wn, D, device="cuda"), torch.randn(B * Wn, D, device="cuda")
        outs = {}
        for mode in (2, 0):
            tune(mode)
            outs[mode] = ops.attention(q, lk, lv, B, H, NT, Wn, 1, (NT, 0, 1), (Wn, 0, 1))
        torch.cuda.synchronize()
        ref = torch.softmax((q.view(B, NT, H, 128).permute(0, 2, 1, 3).double() @ lk.view(B, Wn, H, 128).permute(0, 2, 3, 1).double()) / 128 ** 0.5, -1) @ lv.view(B, Wn, H, 128).permute(0, 2, 1, 3).double()
        ref = ref.permute(0, 2, 1, 3).reshape(B * NT, D)
        print(f"o2l B={B} Sq={NT} Sk={Wn}: ring vs res max|diff| {float((outs[2] - outs[0]).abs().max()):.3e}; vs float64: ring {float((outs[2] - ref).abs().max()):.3e} res {float((outs[0] - ref).abs().max()):.3e}", flush=True)
    tune(1)

if what in ("time", "all"):
    Wn = 48
    only = sys.argv[2] if len(sys.argv) > 2 else None
    modes = [int(c) for c in sys.argv[3]] if len(sys.argv) > 3 else [0, 1, 2]
    for tag, B, N, Tp in (("NS", 256, 64, 4), ("N80", 256, 80, 4), ("C4", 32, 128, 16), ("N16", 512, 16, 4)):
        if only and tag != only:
            continue
        M = B * N * Tp
        q, k, v = (torch.randn(M, D, device="cuda") for _ in range(3))
        lk, lv = torch.randn(B * Wn, D, device="cuda"), torch.randn(B * Wn, D, device="cuda")
        cases = {
            f"obj Sq=Sk={N}": (lambda: ops.attention(q, k, v, B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp)), 4 * M * D * 4),
            f"o2l Sq={N * Tp} Sk=48": (lambda: ops.attention(q, lk, lv, B, H, N * Tp, Wn, 1, (N * Tp, 0, 1), (Wn, 0, 1)), (2 * M + 2 * B * Wn) * D * 4),
        }
        for mode in modes:
            tune(mode)
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
                line.append(f"{name}: {best * 1e3:6.1f} us {nbytes / best / 1e6 / 8000 * 100:4.1f}%")
            print(f"{tag:4s} attn_ring={mode} " + " | ".join(line), flush=True)
        del q, k, v, lk, lv
    tune(1)
