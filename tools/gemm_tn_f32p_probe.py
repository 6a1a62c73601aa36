"""Round 5: the persistent direct-to-LDS exact-f32 weight-gradient GEMM (gemm_tn_f32p.hip) against the one-tile kernel (sola_tune "gemm_tn_persist" 0)
and a float64 product: errors of both, launch times.  Also the whole exact-f32 ragged training step with the key on / off."""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
PEAK = 157.3


def t(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ab(fn):
    out = {}
    for v in (0, 1):
        _lib.check(lib.sola_tune(b"gemm_tn_persist", v), "tune")
        y = fn(); torch.cuda.synchronize()
        out[v] = (t(fn), y)
    return out


torch.manual_seed(0)
bad = 0
for (M, N, K, wb) in [(65536, 1024, 1024, True), (40930, 1024, 1024, True), (65536, 1024, 3072, False), (40930, 1024, 1024, False), (16384, 512, 1024, True), (5000, 1024, 1024, True)]:
    a = torch.randn(M, N, device="cuda"); b = torch.randn(M, K, device="cuda")
    o = ab(lambda: ops.gemm_tn(a, b, wb))
    ref = (a[:, :64].double().t() @ b.double()).float()  # first 64 output rows in float64
    y0 = o[0][1][0] if wb else o[0][1]; y1 = o[1][1][0] if wb else o[1][1]
    e0 = float((y0[:64] - ref).abs().max()); e1 = float((y1[:64] - ref).abs().max()); d = float((y0 - y1).abs().max())
    ok = e1 <= max(2.0 * e0, 1e-3 * float(ref.abs().max())) and d <= 4.0 * max(e0, e1) + 1e-6
    msg = ""
    if wb:
        rb = a.double().sum(0).float()
        eb0 = float((o[0][1][1] - rb).abs().max()); eb1 = float((o[1][1][1] - rb).abs().max())
        ok = ok and eb1 <= max(2.0 * eb0, 1e-3 * float(rb.abs().max()))
        msg = f" bias err {eb0:.2e} -> {eb1:.2e}"
    bad += not ok
    fl = 2 * M * N * K / 1e6
    print(f"tn M={M} N={N} K={K}: one-tile {o[0][0]:.0f} us ({fl / o[0][0] / PEAK:.3f}) -> persistent {o[1][0]:.0f} us ({fl / o[1][0] / PEAK:.3f} of peak);"
          f" err vs f64 {e0:.2e} -> {e1:.2e}, max diff {d:.2e}{msg} {'ok' if ok else 'BAD'}", flush=True)
    del a, b, o
for (R, T, cin, cout, k, st, pd) in [(4096, 32, 256, 512, 3, 2, 1), (8192, 8, 512, 512, 3, 2, 1), (16384, 4, 512, 1024, 3, 1, 1), (16384, 4, 1024, 1024, 1, 1, 0), (3001, 33, 256, 512, 3, 2, 1)]:
    x = torch.randn(R, T, cin, device="cuda"); w = torch.randn(cout, k * cin, device="cuda") * 0.03
    tout = (T + 2 * pd - k) // st + 1
    dy = torch.randn(R, tout, cout, device="cuda")
    o = ab(lambda: ops.conv1d_cl_backward(x, w, dy, k, st, pd, need_dx=False))
    # float64 reference of dW: unfold the windows
    xp = torch.nn.functional.pad(x.double().permute(0, 2, 1), (pd, pd))           # [R, cin, T + 2 pd]
    cols = xp.unfold(2, k, st).permute(0, 2, 3, 1).reshape(R * tout, k * cin)      # [R * tout, k * cin] with column = tap * cin + ci
    ref = (dy.double().reshape(R * tout, cout)[:, :32].t() @ cols).float()
    e0 = float((o[0][1][1][:32] - ref).abs().max()); e1 = float((o[1][1][1][:32] - ref).abs().max()); d = float((o[0][1][1] - o[1][1][1]).abs().max())
    db = float((o[0][1][2] - o[1][1][2]).abs().max())
    ok = e1 <= max(2.0 * e0, 1e-3 * float(ref.abs().max())) and db <= 1e-3 * float(o[0][1][2].abs().max())
    bad += not ok
    fl = 2 * R * tout * cout * k * cin / 1e6
    print(f"conv dW R={R} T={T} cin={cin} cout={cout} k={k} s={st}: one-tile {o[0][0]:.0f} us -> persistent {o[1][0]:.0f} us (whole backward call, no dX; {fl / 1e6:.2f} TFLOP of dW);"
          f" err vs f64 {e0:.2e} -> {e1:.2e}, max diff {d:.2e}, db diff {db:.2e} {'ok' if ok else 'BAD'}", flush=True)
    del x, w, dy, o
print("FAILURES", bad)
_lib.check(lib.sola_tune(b"gemm_tn_persist", 1), "tune")
sys.exit(1 if bad else 0)
