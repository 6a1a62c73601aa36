"""Split-f16 q / k / v attention: the high-occupancy shape for split inputs (attn_fwd_spin_kernel, sola_tune attn_spin 1) against
attn.hip's round-1 kernel for split inputs (0) and the exact-f32 shapes on f32 inputs, per attention of an alignment layer.
Time per launch (in-library HIP events), fraction of the HBM peak on the algorithmic bytes, largest difference from the f32 result.

    python tools/attn_spin_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, ops  # noqa: E402

lib = _lib.lib()
D, H, Wn = 1024, 8, 48


def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        _lib.profile_enable(True); _lib.profile_read(reset=True)
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        best = min(best, _lib.profile_read(reset=True)["attn"]["ms"] / 10)
        _lib.profile_enable(False)
    return best


for tag, B, N, Tp in (("NS", 256, 64, 4), ("N80", 192, 80, 4), ("C4", 32, 128, 16)):
    M = B * N * Tp
    q, k, v = (torch.randn(M, D, device="cuda") for _ in range(3))
    lk, lv = torch.randn(B * Wn, D, device="cuda"), torch.randn(B * Wn, D, device="cuda")
    qs, ks, vs, lks, lvs = (ops.cast_sp16(t) for t in (q, k, v, lk, lv))
    cases = {
        f"obj {N}x{N}": ((q, k, v), (qs, ks, vs), (B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp)), 4 * M * D * 4),
        f"o2l {N * Tp}x48": ((q, lk, lv), (qs, lks, lvs), (B, H, N * Tp, Wn, 1, (N * Tp, 0, 1), (Wn, 0, 1)), (2 * M + 2 * B * Wn) * D * 4),
    }
    if Tp > 4:
        cases[f"motion {Tp}x{Tp}"] = ((q, k, v), (qs, ks, vs), (B * N, H, Tp, Tp, 1, (Tp, 0, 1), (Tp, 0, 1)), 4 * M * D * 4)
    for name, (f32in, spin, geo, nbytes) in cases.items():
        ref = ops.attention(*f32in, *geo)
        t32 = timed(lambda: ops.attention(*f32in, *geo))
        out = {}
        for sw in (1, 2, 0):  # double-buffered 16-key stages (default), single-buffered 32-key stages, attn.hip's kernel
            _lib.check(lib.sola_tune(b"attn_spin", sw), "tune")
            try:
                o = ops.attention_split(*spin, *geo)
                out[sw] = (timed(lambda: ops.attention_split(*spin, *geo)), float((o - ref).abs().max()))
            except Exception as e:  # noqa: BLE001 - the round-1 kernel does not take every shape
                out[sw] = (float("nan"), float("nan"))
        _lib.check(lib.sola_tune(b"attn_spin", 1), "tune")
        print(f"{tag:4s} {name:16s} f32 shapes {t32 * 1e3:6.1f} us ({nbytes / t32 / 1e6 / 8000 * 100:4.1f}%) | split inputs: new {out[1][0] * 1e3:6.1f} us "
              f"({nbytes / out[1][0] / 1e6 / 8000 * 100:4.1f}%) maxdiff {out[1][1]:.1e} [single-buffered {out[2][0] * 1e3:6.1f}] | round-1 kernel {out[0][0] * 1e3:6.1f} us maxdiff {out[0][1]:.1e}", flush=True)
