"""Where the split-f16 GEMM's time goes: the same launch with pieces switched off (sola_tune "gemm_ablate":
1 = no DMA after the first tiles, 2 = no MFMA/fragment reads, 4 = no epilogue)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
for (M, N, K) in [(16384, 1024, 1024), (16384, 1024, 3072), (65536, 512, 768)]:
    x = torch.randn(M, K, device="cuda"); wt = torch.randn(N, K, device="cuda") * 0.03
    a = ops.cast_sp16(x); w = ops.cast_sp16(wt, 64.0); b = torch.randn(N, device="cuda"); r = ops.cast_sp16(torch.randn(M, N, device="cuda"))
    for v in (1, 4):
        lib.sola_tune(b"gemm_glds", v)
        row = []
        for ab in (0, 1, 4, 5):
            lib.sola_tune(b"gemm_ablate", ab)
            best = 1e9
            for rnd in range(3):
                ops.gemm_nt_split(a, w, b, r, True, 1 / 64); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): ops.gemm_nt_split(a, w, b, r, True, 1 / 64)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            row.append(f"{ {0:'full',1:'noDMA',4:'noEpi',5:'compute-only'}[ab]}: {best*1e3:.1f}")
        print(f"M={M} N={N} K={K} glds{128 if v == 1 else 256}x: " + "  ".join(row) + " us")
lib.sola_tune(b"gemm_ablate", 0); lib.sola_tune(b"gemm_glds", 3)
