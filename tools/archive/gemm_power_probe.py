"""Is the split GEMM power-bound?  Same launches with random and with all-zero operands (zero operands toggle no
matrix-pipe inputs): if the whole kernel speeds up with zeros, the limiter is the chip's power management, not a pipeline
bubble that overlap could hide."""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
NAMES = {0: "full", 1: "noDMA", 4: "noEpi", 5: "compute-only"}
M, N, K = 65536, 1024, 1024
for data in ("randn", "zeros", "small-int"):
    if data == "randn":
        x = torch.randn(M, K, device="cuda"); wt = torch.randn(N, K, device="cuda") * 0.03
    elif data == "zeros":
        x = torch.zeros(M, K, device="cuda"); wt = torch.zeros(N, K, device="cuda")
    else:
        x = torch.randint(0, 2, (M, K), device="cuda").float(); wt = torch.randint(0, 2, (N, K), device="cuda").float() / 64
    a = ops.cast_sp16(x); w = ops.cast_sp16(wt, 64.0); b = torch.randn(N, device="cuda")
    r = ops.cast_sp16(torch.randn(M, N, device="cuda"))
    del x
    for persist in (0, 1):
        lib.sola_tune(b"gemm_persist", persist)
        row = []
        for ab in ((0, 1, 4, 5) if persist == 0 else (0, 4)):
            lib.sola_tune(b"gemm_ablate", ab)
            best = 1e9
            for rnd in range(3):
                ops.gemm_nt_split(a, w, b, None, True, 1 / 64, False); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.gemm_nt_split(a, w, b, None, True, 1 / 64, False)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            row.append(f"{NAMES[ab]}: {best*1e3:.1f}")
        print(f"data={data} persist={persist}: " + "  ".join(row) + " us", flush=True)
lib.sola_tune(b"gemm_ablate", 0); lib.sola_tune(b"gemm_persist", 1)
