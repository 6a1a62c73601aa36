"""Steady-state (multi-round) cost of the split GEMM's epilogue: M=65536 N=K=1024 (the projection shape of the default
bench) with and without residual / split output, full vs no-epilogue vs compute-only."""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
NAMES = {0: "full", 1: "noDMA", 4: "noEpi", 5: "compute-only"}
shapes = [(65536, 1024, 1024), (65536, 1024, 3072), (262144, 512, 768)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1:4])]
for (M, N, K) in shapes:
    x = torch.randn(M, K, device="cuda"); wt = torch.randn(N, K, device="cuda") * 0.03
    a = ops.cast_sp16(x); w = ops.cast_sp16(wt, 64.0); b = torch.randn(N, device="cuda")
    r = ops.cast_sp16(torch.randn(M, N, device="cuda"))
    del x
    for (res, osp, persist) in [(False, False, 0), (False, False, 1), (True, False, 0), (True, False, 1), (True, True, 0), (True, True, 1)]:
        lib.sola_tune(b"gemm_persist", persist)
        row = []
        for ab in (0, 4):
            lib.sola_tune(b"gemm_ablate", ab)
            best = 1e9
            for rnd in range(3):
                ops.gemm_nt_split(a, w, b, r if res else None, True, 1 / 64, osp); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.gemm_nt_split(a, w, b, r if res else None, True, 1 / 64, osp)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            row.append(f"{NAMES[ab]}: {best*1e3:.1f}")
        tf = 6.0 * M * N * K / 1e12
        print(f"M={M} N={N} K={K} residual={int(res)} out_split={int(osp)} persist={persist}: " + "  ".join(row) + f" us   ({tf:.2f} TF executed)", flush=True)
lib.sola_tune(b"gemm_ablate", 0)
