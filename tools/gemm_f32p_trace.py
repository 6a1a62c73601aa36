"""Round 5 measurement (EXPERIMENTS=1 build): per-phase cycle sums of the persistent exact-f32 GEMM's k-loop (sola_tune "gemm_f32p_ablate" 128 / 160)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
lib.sola_gemm_f32p_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
M, N = 65536, 1024
for K in (1024, 3072):
  for mode in (192,):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.03; b = torch.randn(N, device="cuda")
    _lib.check(lib.sola_tune(b"gemm_f32p_ablate", mode), "tune")
    for _ in range(3): ops.gemm_nt(a, w, b, None)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 12, dtype=np.uint64)
    lib.sola_gemm_f32p_trace_read(buf.ctypes.data, buf.size)
    rec = buf[: 256 * 8 * 4].reshape(256, 8, 4).astype(np.float64); ph = buf[256 * 8 * 4:].reshape(256, 8, 8).astype(np.float64)
    nkt = rec[:, :, 3].mean() * (K // 32)
    print(f"K={K} mode={mode}: k-tiles per wave {nkt:.0f}; per k-tile and wave (mean over blocks), clock64 ticks:")
    for wv in range(8):
        p = ph[:, wv, :5].mean(0) / nkt
        print(f"  wave {wv}: steps0-2 {p[0]:.0f}  dma-wait {p[1]:.0f}  barrier {p[2]:.0f}  step3 {p[3]:.0f}  fold {p[4]:.0f}  | sum {p.sum():.0f}  loop/kt {rec[:, wv, 0].mean() / nkt:.0f}  epi/tile {rec[:, wv, 2].mean() / rec[:, wv, 3].mean():.0f}")
_lib.check(lib.sola_tune(b"gemm_f32p_ablate", 0), "tune")
