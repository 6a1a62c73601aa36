"""What the DMA costs the 256x256 loop (zero operands, so the matrix pipe is not power-throttled; no epilogue)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
lib.sola_tune(b"gemm_persist", 0)
M, N, K = 65536, 1024, 1024
for data in ("zeros", "randn"):
    x = torch.zeros(M, K, device="cuda") if data == "zeros" else torch.randn(M, K, device="cuda")
    wt = torch.zeros(N, K, device="cuda") if data == "zeros" else torch.randn(N, K, device="cuda") * 0.03
    a = ops.cast_sp16(x); w = ops.cast_sp16(wt, 64.0); b = torch.randn(N, device="cuda")
    row = []
    for ab, name in ((4, "DMA"), (5, "no DMA"), (6, "DMA from one cached line"), (12, "DMA, no vmcnt wait"), (14, "cached line + no wait")):
        lib.sola_tune(b"gemm_ablate", ab)
        best = 1e9
        for rnd in range(3):
            ops.gemm_nt_split(a, w, b, None, True, 1 / 64, False); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm_nt_split(a, w, b, None, True, 1 / 64, False)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        row.append(f"{name}: {best*1e3:.1f}")
    print(f"data={data} (no epilogue): " + "  ".join(row) + " us", flush=True)
lib.sola_tune(b"gemm_ablate", 0); lib.sola_tune(b"gemm_persist", 1)
