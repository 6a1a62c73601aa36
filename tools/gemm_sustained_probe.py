"""Sustained per-launch time of the persistent split-f16 GEMM on a few shapes, in chunks of launches with rotating operand buffers: does the time drift, is a launch slower inside a step than alone?"""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops
def run(M, N, K, nbuf, iters, chunks):
    As = [ops.cast_sp16(torch.randn(M, K, device="cuda")) for _ in range(nbuf)]
    w = ops.cast_sp16(torch.randn(N, K, device="cuda") * 0.03, 64.0)
    b = torch.randn(N, device="cuda")
    for i in range(3): ops.gemm_nt_split(As[i % nbuf], w, b, None, True, 1 / 64, False)
    torch.cuda.synchronize()
    out = []
    for c in range(chunks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): ops.gemm_nt_split(As[i % nbuf], w, b, None, True, 1 / 64, False)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    print(f"M={M} N={N} K={K} nbuf={nbuf}: us per launch by chunk of {iters}: " + " ".join(f"{t:.0f}" for t in out), flush=True)
run(262144, 256, 256, 1, 50, 12)
run(262144, 256, 256, 4, 50, 12)
run(262144, 768, 256, 1, 20, 12)
run(262144, 768, 256, 4, 20, 12)
run(262144, 256, 768, 4, 20, 8)
run(65536, 1024, 1024, 4, 20, 8)
