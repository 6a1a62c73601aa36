"""Attention backward per shape: block-shared staging (attn_bwd_blk 1) vs the per-wave kernels (0); HIP events.
usage: attn_bwd_probe.py [B N Tp]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
Tp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
D, H, Wn = 1024, 8, 48
M = B * N * Tp
lib = _lib.lib()
q, k, v, do = (torch.randn(M, D, device="cuda") for _ in range(4))
lk, lv = torch.randn(B * Wn, D, device="cuda"), torch.randn(B * Wn, D, device="cuda")
geo = {"obj": (q, k, v, B * Tp, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp)),
       "motion": (q, k, v, B * N, Tp, Tp, 1, (Tp, 0, 1), (Tp, 0, 1)),
       "o2l": (q, lk, lv, B, N * Tp, Wn, 1, (N * Tp, 0, 1), (Wn, 0, 1))}
for name, (qq, kk, vv, G, Sq, Sk, inner, qa, ka) in geo.items():
    o, lse = ops.attention(qq, kk, vv, G, H, Sq, Sk, inner, qa, ka, return_lse=True)
    res, best = {}, {}
    for rnd in range(3):
        for mode in (0, 1):
            lib.sola_tune(b"attn_bwd_blk", mode)
            fn = lambda: ops.attention_backward(qq, kk, vv, o, do, lse, G, H, Sq, Sk, inner, qa, ka)
            res[mode] = fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            best[mode] = min(best.get(mode, 1e9), e0.elapsed_time(e1) / 10)
    lib.sola_tune(b"attn_bwd_blk", 1)
    diff = max(float((a - b).abs().max()) for a, b in zip(res[0], res[1]))
    print(f"{name:8s} Sq={Sq} Sk={Sk}: per-wave {best[0]*1e3:7.1f} us   block-shared {best[1]*1e3:7.1f} us   maxdiff {diff:.1e}")
