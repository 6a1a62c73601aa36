import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sola_amd import ops, _lib
lib = _lib.lib()
for (M, N, K) in [(65536, 1024, 1024), (131072, 512, 768)]:
    x = torch.randn(M, K, device="cuda"); wt = torch.randn(N, K, device="cuda") * 0.03
    a = ops.cast_sp16(x); w = ops.cast_sp16(wt, 64.0); b = torch.randn(N, device="cuda")
    for pp in (0, 1):
        lib.sola_tune(b"gemm_pp", pp)
        row = []
        for ab in (0, 4, 8, 16, 24):
            lib.sola_tune(b"gemm_ablate", ab)
            best = 1e9
            for rnd in range(3):
                ops.gemm_nt_split(a, w, b, out_scale=1 / 64); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.gemm_nt_split(a, w, b, out_scale=1 / 64)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            row.append(best * 1e3)
        print(f"M={M} N={N} K={K} pp={pp}: full {row[0]:7.1f}  no drain {row[1]:7.1f}  no stores {row[2]:7.1f}  no LDS/convert {row[3]:7.1f}  neither {row[4]:7.1f}")
lib.sola_tune(b"gemm_ablate", 0); lib.sola_tune(b"gemm_pp", 0)
