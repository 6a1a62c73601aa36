"""A/B of two sola_tune settings of the split GEMM, interleaved so that clock / power drift hits both alike.
usage: gemm_ab_probe.py key valueA valueB   (default: gemm_persist 0 1)"""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
key = (sys.argv[1] if len(sys.argv) > 1 else "gemm_persist").encode()
va, vb = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 1)
for kv in sys.argv[4:]:  # fixed settings: key=value
    k_, v_ = kv.split("="); lib.sola_tune(k_.encode(), int(v_))
def once(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
import os
SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ.get('SHAPES', '').split(',') if t] or [(65536, 1024, 1024, 0, 0), (65536, 1024, 1024, 0, 1), (65536, 1024, 1024, 1, 0), (65536, 1024, 3072, 0, 0),
                            (262144, 512, 768, 0, 0), (262144, 512, 768, 1, 0), (131072, 256, 768, 0, 0), (65536, 1024, 512, 0, 0)]
for (M, N, K, res, osp) in SHAPES:
    a = ops.cast_sp16(torch.randn(M, K, device="cuda")); w = ops.cast_sp16(torch.randn(N, K, device="cuda") * 0.03, 64.0)
    b = torch.randn(N, device="cuda"); r = ops.cast_sp16(torch.randn(M, N, device="cuda")) if res else None
    fn = lambda: ops.gemm_nt_split(a, w, b, r, True, 1 / 64, bool(osp))
    ta, tb = [], []
    for rnd in range(6):
        lib.sola_tune(key, va); fn(); torch.cuda.synchronize(); ta.append(once(fn))
        lib.sola_tune(key, vb); fn(); torch.cuda.synchronize(); tb.append(once(fn))
    tf = 6.0 * M * N * K / 1e6
    print(f"M={M} N={N} K={K} residual={res} out_split={osp}: {key.decode()}={va}: min {min(ta):.1f} med {sorted(ta)[3]:.1f} us ({tf/min(ta):.0f} TF/s)"
          f"   {key.decode()}={vb}: min {min(tb):.1f} med {sorted(tb)[3]:.1f} us ({tf/min(tb):.0f} TF/s)   B/A {min(tb)/min(ta):.3f}", flush=True)
    del a, w, r
