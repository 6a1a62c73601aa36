"""Attention-core probe: interleaved A/B of the kernel variants at the three layouts of the headline batch, HIP events."""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
for name, (fn, nbytes) in cases.items():
    best, outs = {}, {}
    for rnd in range(3):
        for var in (3, 1):
            lib.sola_tune(b"attn_variant", var)
            outs[var] = fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            best[var] = min(best.get(var, 1e9), e0.elapsed_time(e1) / 20)
    print(f"{name:22s} " + "  ".join(f"v{var}: {best[var]*1e3:7.1f} us {nbytes/best[var]/1e6:7.1f} GB/s ({nbytes/best[var]/1e6/8000*100:4.1f}% of 8 TB/s)" for var in (3, 1)),
          " maxdiff", float((outs[3] - outs[1]).abs().max()), "scale", float(outs[3].abs().max()))
lib.sola_tune(b"attn_variant", 1)
# the split-f16 MFMA shape on q, k, v already stored as split-f16 rows (what the fast forward path runs)
qs, ks, vs = (ops.cast_sp16(t) for t in (q, k, v)); lks, lvs = ops.cast_sp16(lk), ops.cast_sp16(lv)
split_cases = {
    f"obj split (Sq=Sk={N})": (lambda: ops.attention_split(qs, ks, vs, B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp), out_split=True), 4 * M * D * 4),
    f"o2l split (Sq={N * Tp},Sk=48)": (lambda: ops.attention_split(qs, lks, lvs, B, H, N * Tp, Wn, 1, (N * Tp, 0, 1), (Wn, 0, 1), out_split=True), (2 * M + 2 * B * Wn) * D * 4),
}
if N > 16:
    for name, (fn, nbytes) in split_cases.items():
        fn(); torch.cuda.synchronize(); best = 1e9
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print(f"{name:28s} {best*1e3:7.1f} us {nbytes/best/1e6:7.1f} GB/s ({nbytes/best/1e6/8000*100:4.1f}% of 8 TB/s)")
