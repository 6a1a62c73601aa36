"""Round 6: would the backward's weight-gradient product (dW = dY^T X, gemm_tn_tr_kernel + its reduce) overlap usefully with the input-gradient GEMM of the
same dY (dX = dY W, persistent 256x256 kernel: 632 tiles = 2.47 rounds of the 256 CUs at M = 40 320) if it ran on a second stream?  Upper bound: here the dW
call also carries its two operand casts, which the training step does not run (the operands are already bfloat16 there).
    python tools/gemm_overlap_probe.py [M = 40320]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd._lib import check, lib, ptr
import ctypes as C
M = int(sys.argv[1]) if len(sys.argv) > 1 else 40320
N = K = 1024
dev = torch.device("cuda", 0)
dy16 = torch.randn(M, N, device=dev).to(torch.bfloat16); w16 = (torch.randn(K, N, device=dev) / 32).to(torch.bfloat16)
res = torch.randn(M, K, device=dev); dx = torch.empty(M, K, device=dev)
dy32 = torch.randn(M, N, device=dev) * 1e-3; x32 = torch.randn(M, K, device=dev); dw = torch.empty(N, K, device=dev)
nb = lib().sola_gemm_tn_split_scratch_bytes(M, N, K); scratch = torch.empty(nb, device=dev, dtype=torch.uint8)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def dX(st):
    check(lib().sola_gemm_nt_bf16(ptr(dy16), N, ptr(w16), None, ptr(res), K, 0, ptr(dx), K, 0, M, K, N, C.c_void_p(st.cuda_stream)), "dX")
def dW(st):
    check(lib().sola_gemm_tn_f16(ptr(dy32), N, ptr(x32), K, ptr(dw), M, N, K, 2, ptr(scratch), nb, C.c_void_p(st.cuda_stream)), "dW")
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def serial():
    dX(s1); dW(s1)
def overlapped():
    ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
    dX(s1); dW(s2)
    ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
print(f"M={M}: dX alone {timed(lambda: dX(s1)):.1f} us, dW (+ casts) alone {timed(lambda: dW(s1)):.1f} us, serial {timed(serial):.1f} us, two streams {timed(overlapped):.1f} us")
