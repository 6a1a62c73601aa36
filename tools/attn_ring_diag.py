"""Ring attention kernel, inter-object shape (B=256, N=64, T'=4): launches per configuration of sola_tune keys, to be read from a
rocprofv3 kernel trace (tools/prof_stats.sh) - the configurations run in the order given, 30 launches each.

    [RING_DIMS=B,N,Tp] [RING_SHAPE=obj|o2l] python tools/attn_ring_diag.py "attn_ring_blocks=1" "attn_ring_blocks=2" "attn_ring_ablate=1" ...   (EXPERIMENTS=1 build for ablate)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, ops  # noqa: E402

lib = _lib.lib()
D, H = 1024, 8
B, N, Tp = (int(x) for x in os.environ.get("RING_DIMS", "256,64,4").split(","))
shape = os.environ.get("RING_SHAPE", "obj")
M = B * N * Tp
q, k, v = (torch.randn(M, D, device="cuda") for _ in range(3))
Wn = 48
lk, lv = torch.randn(B * Wn, D, device="cuda"), torch.randn(B * Wn, D, device="cuda")
fused = os.environ.get("RING_FUSED") == "1"  # q | k | v as the column blocks of one [M, 3 D] matrix (the projection GEMM's output in the model)
if fused:
    import ctypes as C
    qkv = torch.randn(M, 3 * D, device="cuda")
    out = torch.empty(M, D, device="cuda")
    stream = _lib.current_stream(out.device)

    def fused_obj():
        base = qkv.data_ptr()
        _lib.check(lib.sola_attention(C.c_void_p(base), 3 * D, C.c_void_p(base + 4 * D), 3 * D, C.c_void_p(base + 8 * D), 3 * D, _lib.ptr(out), D,
                                      B * Tp, H, D // H, N, N, Tp, N * Tp, 1, Tp, N * Tp, 1, Tp, 1.0 / (D // H) ** 0.5, None, stream), "sola_attention")
defaults = {"attn_ring": 1, "attn_ring_blocks": 2, "attn_ring_remap": 1} if lib.sola_has_experiments() else {}
for cfg in sys.argv[1:]:
    kv = dict(defaults)
    for item in cfg.split(","):
        key, val = item.split("=")
        kv[key] = int(val)
    if lib.sola_has_experiments():
        kv.setdefault("attn_ring_ablate", 0)
    for key, val in kv.items():
        _lib.check(lib.sola_tune(key.encode(), val), key)
    for _ in range(30):
        if fused:
            fused_obj()
        elif shape == "obj":
            ops.attention(q, k, v, B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp))
        else:
            ops.attention(q, lk, lv, B, H, N * Tp, Wn, 1, (N * Tp, 0, 1), (Wn, 0, 1))
    torch.cuda.synchronize()
    print("ran", cfg, flush=True)
