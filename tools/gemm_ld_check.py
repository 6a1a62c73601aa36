"""Experiment "gemm_ld" (one wave of each SIMD's pair issues the whole DMA stream, buffer loads): bit-identity against the default persistent
kernel (interior and ragged-edge shapes), then interleaved timing.  usage: gemm_ld_check.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
def run(a, w, b, ld):
    lib.sola_tune(b"gemm_ld", ld)
    out = ops.gemm_nt_split(a, w, b, None, True, 1 / 64, False)
    torch.cuda.synchronize()
    return out
torch.manual_seed(0)
for (M, N, K) in [(65536, 1024, 1024), (65536 + 100, 1000, 1024), (32768 + 8, 2048, 768), (131072, 512, 768)]:
    a = ops.cast_sp16(torch.randn(M, K, device="cuda")); w = ops.cast_sp16(torch.randn(N, K, device="cuda") * 0.03, 64.0)
    b = torch.randn(N, device="cuda")
    ref = run(a, w, b, 0)
    for ld in (1, 2):
        o = run(a, w, b, ld)
        print(f"M={M} N={N} K={K} gemm_ld={ld}: bit-identical {torch.equal(o, ref)}  max|diff| {float((o - ref).abs().max()):.3e}  finite {bool(torch.isfinite(o).all())}", flush=True)
    del a, w, ref
lib.sola_tune(b"gemm_ld", 0)
