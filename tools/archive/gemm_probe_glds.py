"""Split-f16 GEMM: register-staged (0) vs direct-to-LDS 128x128 / 2 stages (1) vs 256x256 blocks of 128x64 wave tiles (4), interleaved
A/B/C with result comparison."""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
NAMES = {0: 'regs', 1: 'glds128', 4: 'glds256x256'}
for (M, N, K) in [(16384, 1024, 1024), (16384, 1024, 3072), (65536, 512, 768), (16384, 512, 1536), (8192, 1024, 1024), (16000, 1000, 992), (16384, 1024, 32), (300, 1024, 64)]:
    x = torch.randn(M, K, device="cuda"); wt = torch.randn(N, K, device="cuda") * 0.03
    a = ops.cast_sp16(x); w = ops.cast_sp16(wt, 64.0); b = torch.randn(N, device="cuda"); r = ops.cast_sp16(torch.randn(M, (N // 8) * 8, device="cuda")) if N % 8 == 0 else None
    best, outs = {}, {}
    for rnd in range(3):
        for v in (0, 1, 4):
            lib.sola_tune(b"gemm_glds", v)
            outs[v] = ops.gemm_nt_split(a, w, b, r, r is not None, 1 / 64); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.gemm_nt_split(a, w, b, r, r is not None, 1 / 64)
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best.get(v, 1e9), e0.elapsed_time(e1) / 20)
    print(f"M={M} N={N} K={K}: " + "  ".join(f"{NAMES[v]}: {best[v]*1e3:.1f} us {2*M*N*K/best[v]/1e9:.0f} TF-alg ({3*2*M*N*K/best[v]/1e9/2500*100:.1f}% f16 peak)" for v in (0, 1, 4)),
          " maxdiff", float((outs[0] - outs[1]).abs().max()), float((outs[0] - outs[4]).abs().max()), "ref-scale", float(outs[0].abs().max()))
lib.sola_tune(b"gemm_glds", 3)
