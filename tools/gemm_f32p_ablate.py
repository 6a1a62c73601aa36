"""Round 5 measurement (EXPERIMENTS=1 build): where the persistent exact-f32 GEMM's time goes - ablations (results are garbage) and cycle stamps."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(65536, 1024, 1024), (65536, 1024, 3072)]:
    for zeros in (False,):
        a = torch.zeros(M, K, device="cuda") if zeros else torch.randn(M, K, device="cuda")
        w = torch.zeros(N, K, device="cuda") if zeros else torch.randn(N, K, device="cuda") * 0.03
        r = torch.randn(M, N, device="cuda"); b = torch.randn(N, device="cuda")
        fl = 2 * M * N * K / 1e6
        for res in (None, r):
            line = f"M={M} K={K} zeros={zeros} res={res is not None}:"
            for abl in (0, 64, 0, 64):
                _lib.check(lib.sola_tune(b"gemm_f32p_ablate", abl), "tune")
                us = t(lambda: ops.gemm_nt(a, w, b, res))
                line += f"  abl{abl} {us:.0f}us({fl / us / 157.3:.3f})"
            print(line, flush=True)
        del a, w, r
M, N, K = 65536, 1024, 1024
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.03; b = torch.randn(N, device="cuda")
_lib.check(lib.sola_tune(b"gemm_f32p_ablate", int(sys.argv[1]) if len(sys.argv) > 1 else 16), "tune")
for _ in range(3): ops.gemm_nt(a, w, b, None)
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 4, dtype=np.uint64)
lib.sola_gemm_f32p_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.sola_gemm_f32p_trace_read(buf.ctypes.data, buf.size)
rec = buf.reshape(256, 8, 4).astype(np.float64)
print("trace rc", rc, "tiles per block", rec[:, :, 3].mean())
print("per wave (mean over blocks): loop cycles, barrier-wait cycles, epilogue cycles")
for wv in range(8):
    print(f"  wave {wv}: loop {rec[:, wv, 0].mean():.0f}  wait {rec[:, wv, 1].mean():.0f} ({rec[:, wv, 1].mean() / rec[:, wv, 0].mean():.3f})  epi {rec[:, wv, 2].mean():.0f}")
_lib.check(lib.sola_tune(b"gemm_f32p_ablate", 0), "tune")
