"""GEMM-only probe: interleaved A/B of the GEMM schedule variants at the path's shapes, HIP-event timing."""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
shapes = [(16384, 1024, 1024), (16384, 1024, 3072), (65536, 512, 768), (16384, 512, 1536), (4096, 1024, 1024), (256, 1024, 1024)]
if len(sys.argv) > 1 and sys.argv[1] != "-":
    shapes = [tuple(int(v) for v in sys.argv[1].split("x"))]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1]
lib = _lib.lib()
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.03; b = torch.randn(N, device="cuda")
    best = {v: 1e9 for v in variants}
    outs = {}
    for rnd in range(3):
        for v in variants:
            lib.sola_tune(b"gemm_variant", v)
            outs[v] = ops.gemm_nt(a, w, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): ops.gemm_nt(a, w, b)
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / reps)
    same = all(torch.equal(outs[variants[0]], outs[v]) for v in variants)
    print(f"M={M} N={N} K={K}: " + "  ".join(f"v{v}: {best[v]*1e3:.1f} us {2*M*N*K/best[v]/1e9:.1f} TF ({2*M*N*K/best[v]/1e9/157.3*100:.1f}%)" for v in variants) + f"  identical={same}")
lib.sola_tune(b"gemm_variant", 1)
