"""Per-shape attention time: exact-f32 auto routing vs split-math kernel (f32 inputs), HIP events, standalone."""
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
    f"o2l (Sq={N * Tp},Sk=48)": (lambda: ops.attention(q, lk, lv, B, H, N * Tp, Wn, 1, (N * Tp, 0, 1), (Wn, 0, 1)), (2 * M + 2 * B * Wn) * D * 4),
}
for name, (fn, nbytes) in cases.items():
    best = {}
    for rnd in range(3):
        for sm in (0, 1):
            lib.sola_tune(b"attn_stage_split_math", sm)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            best[sm] = min(best.get(sm, 1e9), e0.elapsed_time(e1) / 20)
    lib.sola_tune(b"attn_stage_split_math", 0)
    print(f"{name:22s} " + "  ".join(f"split_math={sm}: {best[sm]*1e3:7.1f} us ({nbytes/best[sm]/1e6/8000*100:4.1f}% of 8 TB/s)" for sm in (0, 1)))
