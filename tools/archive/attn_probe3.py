"""Per-shape attention time: register-only shape (attn_reg.hip) vs the LDS shapes, HIP events, standalone.
usage: attn_probe3.py [B N Tp]   (headline: 256 64 4; C4 stress: 32 128 16)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
Tp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
D, H, Wn = 1024, 8, 48
M = B * N * Tp
lib = _lib.lib()
q, k, v = (torch.randn(M, D, device="cuda") for _ in range(3))
lk, lv = torch.randn(B * Wn, D, device="cuda"), torch.randn(B * Wn, D, device="cuda")
cases = {
    f"obj (Sq=Sk={N})": (lambda: ops.attention(q, k, v, B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp)), 4 * M * D * 4),
    f"motion (Sq=Sk={Tp})": (lambda: ops.attention(q, k, v, B * N, H, Tp, Tp, 1, (Tp, 0, 1), (Tp, 0, 1)), 4 * M * D * 4),
    f"o2l (Sq={N * Tp},Sk=48)": (lambda: ops.attention(q, lk, lv, B, H, N * Tp, Wn, 1, (N * Tp, 0, 1), (Wn, 0, 1)), (2 * M + 2 * B * Wn) * D * 4),
}
modes = [("lds", 0, 0, 0), ("res8", 0, 1, 1), ("res4 prefetch", 0, 1, 2)]
for name, (fn, nbytes) in cases.items():
    best, outs = {}, {}
    for rnd in range(3):
        for mname, reg, minw, shape in modes:
            lib.sola_tune(b"attn_reg", reg); lib.sola_tune(b"attn_res", 1 if minw else 0); lib.sola_tune(b"attn_res_tiles", 0 if minw <= 1 else minw); lib.sola_tune(b"attn_res_shape", shape)
            o = fn(); torch.cuda.synchronize()
            outs[mname] = o
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            best[mname] = min(best.get(mname, 1e9), e0.elapsed_time(e1) / 20)
    lib.sola_tune(b"attn_reg", 1); lib.sola_tune(b"attn_res", 1); lib.sola_tune(b"attn_res_tiles", 0); lib.sola_tune(b"attn_res_shape", 0)
    diff = max(float((outs[m] - outs[modes[0][0]]).abs().max()) for m, _, _, _ in modes)
    print(f"{name:24s} " + "  ".join(f"{m}: {best[m]*1e3:7.1f} us ({nbytes/best[m]/1e6/8000*100:4.1f}%)" for m, _, _, _ in modes) + f"  maxdiff {diff:.1e}")
