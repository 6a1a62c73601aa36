import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
for B in (64, 16, 256):
    N, Tp, D, H, Wn = 64, 4, 1024, 8, 48
    M = B * N * Tp
    q = torch.randn(M, D, device="cuda"); lk, lv = torch.randn(B * Wn, D, device="cuda"), torch.randn(B * Wn, D, device="cuda")
    fn = lambda: ops.attention(q, lk, lv, B, H, N * Tp, Wn, 1, (N * Tp, 0, 1), (Wn, 0, 1))
    nbytes = (2 * M + 2 * B * Wn) * D * 4
    res = {}
    for rnd in range(3):
        for tb in (256, 384, 512, 640, 768, 1024, 1536, 2048):
            lib.sola_tune(b"attn_target_blocks", tb)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            res[tb] = min(res.get(tb, 1e9), e0.elapsed_time(e1) / 20)
    print(f"B={B} o2l: " + "  ".join(f"{tb}:{t*1e3:.1f}us" for tb, t in res.items()), f" best {nbytes/min(res.values())/1e6:.0f} GB/s")
