"""Is the epilogue of the 256x256 split GEMM bound per CU or chip-wide?  One round of blocks on 32..256 CUs."""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
lib.sola_tune(b"gemm_glds", 4)
def timeit(fn, n=20):
    best = 1e9
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3
N = K = 1024
for M in (2048, 4096, 8192, 16384):
    a = ops.cast_sp16(torch.randn(M, K, device="cuda")); w = ops.cast_sp16(torch.randn(N, K, device="cuda") * 0.03, 64.0)
    b = torch.randn(N, device="cuda"); r = ops.cast_sp16(torch.randn(M, N, device="cuda"))
    for res in (False, True):
        row = []
        for ab in (0, 4):
            lib.sola_tune(b"gemm_ablate", ab)
            row.append(timeit(lambda: ops.gemm_nt_split(a, w, b, r if res else None, True, 1 / 64, False)))
        print(f"blocks={M//256*4} residual={int(res)}: full {row[0]:.1f} us  no-epilogue {row[1]:.1f} us  epilogue {row[0]-row[1]:.1f} us", flush=True)
lib.sola_tune(b"gemm_ablate", 0); lib.sola_tune(b"gemm_glds", 3)
