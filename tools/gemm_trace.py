"""Where the persistent split-f16 GEMM's time goes, from inside the kernel (sola_tune "gemm_trace" 1: instrumented instantiation of the plain
launch - no conv, no residual, f32 output).  Per wave: cycles parked at the k-tile wait + barrier, cycles in the k-loops and in the epilogues;
per tile: 100 MHz stamps of the end of its k-loop and of its epilogue, from which the skew between the CUs' epilogues follows.
usage: gemm_trace.py [MxNxK ...] [key=value ...]   (key=value: further sola_tune settings, e.g. gemm_stagger=40 gemm_ablate=4)"""
import sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
WORDS = 8 + 128
shapes, tunes = [], []
for t in sys.argv[1:]:
    (tunes if "=" in t else shapes).append(t)
shapes = [tuple(int(v) for v in t.split("x")) for t in shapes] or [(65536, 1024, 1024), (65536, 1024, 3072), (65536, 3072, 1024)]
for kv in tunes:
    k, v = kv.split("="); lib.sola_tune(k.encode(), int(v))
for (M, N, K) in shapes:
    a = ops.cast_sp16(torch.randn(M, K, device="cuda")); w = ops.cast_sp16(torch.randn(N, K, device="cuda") * 0.03, 64.0)
    b = torch.randn(N, device="cuda")
    fn = lambda: ops.gemm_nt_split(a, w, b, None, True, 1 / 64, False)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    t_plain = e0.elapsed_time(e1) / 5 * 1e3
    lib.sola_tune(b"gemm_trace", 1)
    fn(); torch.cuda.synchronize()
    e0.record(); [fn() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
    t_trace = e0.elapsed_time(e1) / 5 * 1e3
    buf = np.zeros(1024 * 8 * WORDS, dtype=np.uint64)
    n = lib.sola_gemm_trace_read(buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    lib.sola_tune(b"gemm_trace", 0)
    rec = buf.reshape(1024, 8, WORDS)
    used = rec[:, 0, 3] > 0
    r = rec[used]  # [blocks, 8 waves, WORDS]
    nb = r.shape[0]
    wait, loop, epi, ntile, nk = r[:, :, 0].astype(np.float64), r[:, :, 1].astype(np.float64), r[:, :, 2].astype(np.float64), r[:, :, 3], r[:, :, 6]
    nt = int(ntile[0, 0]); nkt = int(nk[0, 0])
    first, last = r[:, :, 4].astype(np.int64), r[:, :, 5].astype(np.int64)
    kend = r[:, 0, 8:8 + nt].astype(np.int64); eend = r[:, 0, 72:72 + nt].astype(np.int64)
    lo32 = first[:, 0] & 0xffffffff
    d = lambda x, base: ((x - base[:, None]) & 0xffffffff).astype(np.float64) / 100.0  # us since the block's first stamp
    kend_us, eend_us = d(kend, lo32), d(eend, lo32)
    start_us = (first[:, 0] - first[:, 0].min()) / 100.0
    tot = (last - first).astype(np.float64) / 100.0
    print(f"== M={M} N={N} K={K}: {t_plain:.1f} us plain, {t_trace:.1f} us traced; {nb} blocks x {nt} tiles x {nkt} k-tiles; bytes read {n}")
    print(f"   per wave, cycles: k-loops {loop.mean():.0f}  (of which parked at the k-tile wait + barrier {wait.mean():.0f} = {100 * wait.mean() / loop.mean():.1f} %)  epilogues {epi.mean():.0f}"
          f" = {100 * epi.mean() / (loop.mean() + epi.mean()):.1f} % of the wave's time")
    print(f"   per k-tile: {loop.mean() / nt / nkt:.0f} cycles, parked {wait.mean() / nt / nkt:.0f};  per tile: epilogue {epi.mean() / nt:.0f} cycles")
    print(f"   wait share by wave index (mean over blocks): " + " ".join(f"{100 * wait[:, w].mean() / loop[:, w].mean():.1f}" for w in range(8)))
    print(f"   block start spread {start_us.max():.1f} us; block lifetime mean {tot.mean():.1f} us (min {tot.min():.1f} max {tot.max():.1f})")
    for t in range(nt):
        ke = kend_us[:, t] + start_us; ee = eend_us[:, t] + start_us
        print(f"   tile {t}: k-loop ends at {ke.mean():7.1f} us (std {ke.std():5.2f}, min {ke.min():7.1f} max {ke.max():7.1f});  epilogue takes {np.mean(ee - ke):5.2f} us (min {np.min(ee - ke):5.2f} max {np.max(ee - ke):5.2f})")
    del a, w
