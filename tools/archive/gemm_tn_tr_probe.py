"""Weight-gradient GEMM on 16-bit operands: the row-major route (gemm_tn_tr_kernel: transposing LDS reads, sola_tune train_tn_tr 1)
against the transposed-copy route (0): error of both against an f64 product of the rounded operands, and time per call."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sola_amd import ops, _lib
lib = _lib.lib()
torch.manual_seed(0)
def t_us(fn, n=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
bad = 0
for (M, N, K) in [(65536, 1024, 1024), (53211, 1024, 1024), (131072, 512, 768), (65536, 3072, 1024), (65536, 1024, 3072), (16384, 256, 256), (4100, 256, 512), (200, 256, 256)]:
    for bf in (False, True):
        a = torch.randn(M, N, device="cuda") * 1e-4; b = torch.randn(M, K, device="cuda")
        dt = torch.bfloat16 if bf else torch.float16
        amax = float(a.abs().max()); sc = 2.0 ** (13 - (torch.tensor(amax).log2().floor().item()))
        ref = ((a * sc).to(dt).double().t() @ b.to(dt).double()) / sc
        row = []
        for v in (0, 1):
            lib.sola_tune(b"train_tn_tr", v)
            out = ops.gemm_tn_f16(a, b, bf)
            err = float((out.double() - ref).abs().max() / ref.abs().max())
            us = t_us(lambda: ops.gemm_tn_f16(a, b, bf))
            row.append((err, us))
        ok = row[1][0] < 2e-5
        bad += not ok
        print(f"M={M} N={N} K={K} {'bf16' if bf else 'f16 '}: copies err {row[0][0]:.2e} {row[0][1]:8.1f} us   row-major err {row[1][0]:.2e} {row[1][1]:8.1f} us   {'ok' if ok else 'BAD'}", flush=True)
        del a, b, ref
lib.sola_tune(b"train_tn_tr", 1)
sys.exit(1 if bad else 0)
