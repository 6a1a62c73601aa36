"""First-round phase stagger of the 256x256 split GEMM: launch time vs number of phases and assumed k-tile time."""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
def timeit(fn, n=10):
    best = 1e9
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3
for (M, N, K, nprob) in [(65536, 1024, 1024, 1), (65536, 1024, 1024, 3), (65536, 1024, 3072, 1), (262144, 512, 768, 1)]:
    x = torch.randn(M, K, device="cuda"); wt = torch.randn(N, K, device="cuda") * 0.03
    a = ops.cast_sp16(x); w = ops.cast_sp16(wt, 64.0); b = torch.randn(N, device="cuda")
    r = ops.cast_sp16(torch.randn(M, N, device="cuda"))
    del x
    for res in (False, True):
        def fn():
            for _ in range(nprob): ops.gemm_nt_split(a, w, b, r if res else None, True, 1 / 64, False)
        row = []
        for ph in (0, 2, 4):
            for ns in ((2200,) if ph == 0 else (1500, 2200, 3000)):
                lib.sola_tune(b"gemm_stagger", ph); lib.sola_tune(b"gemm_stagger_ns", ns)
                row.append(f"p{ph}/{ns}: {timeit(fn)/nprob:.1f}")
        print(f"M={M} N={N} K={K} x{nprob} residual={int(res)}: " + "  ".join(row) + " us", flush=True)
lib.sola_tune(b"gemm_stagger", 0)
