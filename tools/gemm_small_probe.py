"""Exact-f32 GEMMs of few rows (one sample per call / step): the 32x32 in-block split-K shape (gemm_nt_f32_small_kernel) against the 64x64 +
split-K + reduce pair it replaces, per launch.  usage: gemm_small_probe.py [MxNxK ...]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1:]] or [(256, 1024, 1024), (256, 1024, 3072), (640, 1024, 1024), (1024, 1024, 1024), (2048, 1024, 1024), (256, 512, 768), (48, 2048, 1024)]
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.03; b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda")
    out = {}
    for rows in (0, 4096):
        lib.sola_tune(b"gemm_small_rows", rows)
        out[rows] = t(lambda: ops.gemm_nt(a, w, b, r))
    lib.sola_tune(b"gemm_small_rows", 2048)
    print(f"M={M} N={N} K={K}: 64x64 + split-K {out[0]:.1f} us -> 32x32 in-block {out[4096]:.1f} us  ({2.0 * M * N * K / out[4096] / 1e6:.1f} TFLOP/s)", flush=True)
