"""Round 5: the persistent direct-to-LDS exact-f32 GEMM (gemm_f32p.hip) against the 128x128 one-tile kernel (sola_tune "gemm_f32_persist" 0):
bit-identity of the results (plain, bias, residual, ragged M / N edges, conv windows at stride 1 and 2, the whole exact-f32 forward) and the
launch times.  python tools/gemm_f32p_probe.py [quick]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib, synth
lib = _lib.lib()
PEAK = 157.3


def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ab(fn):
    out = {}
    for v in (0, 1):
        _lib.check(lib.sola_tune(b"gemm_f32_persist", v), "tune")
        y = fn(); torch.cuda.synchronize()
        out[v] = (t(fn), y)
    return out


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
torch.manual_seed(0)
bad = 0
shapes = [(65536, 1024, 1024, True, True), (65536, 1024, 1024, True, False), (65536, 1024, 3072, True, True), (40930, 1024, 1024, True, True),
          (16384, 1024, 1024, False, False), (12288, 1024, 1024, True, False), (65536, 1000, 1024, True, True), (33000, 520, 256, True, True),
          (65536, 512, 64, False, False)]
for (M, N, K, has_b, has_r) in shapes[: 4 if quick else None]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.03
    b = torch.randn(N, device="cuda") if has_b else None
    r = torch.randn(M, N, device="cuda") if has_r else None
    o = ab(lambda: ops.gemm_nt(a, w, b, r))
    same = torch.equal(o[0][1], o[1][1]); bad += not same
    fl = 2 * M * N * K / 1e6
    print(f"gemm M={M} N={N} K={K} bias={has_b} res={has_r}: one-tile {o[0][0]:.0f} us ({fl / o[0][0] / PEAK:.3f}) -> persistent {o[1][0]:.0f} us "
          f"({fl / o[1][0]:.1f} TF = {fl / o[1][0] / PEAK:.3f} of peak); bit-identical {same}", flush=True)
    del a, w, b, r, o
convs = [(4096, 32, 256, 512, 3, 2, 1), (4096, 16, 512, 512, 3, 2, 1), (8192, 8, 512, 512, 3, 2, 1), (16384, 4, 512, 1024, 3, 1, 1),
         (16384, 4, 1024, 1024, 3, 1, 1), (16384, 4, 1024, 1024, 1, 1, 0), (3001, 7, 256, 512, 3, 2, 1)]
for (R, T, cin, cout, k, st, pd) in convs[: 2 if quick else None]:
    x = torch.randn(R, T, cin, device="cuda"); w = torch.randn(cout, k * cin, device="cuda") * 0.03; b = torch.randn(cout, device="cuda")
    o = ab(lambda: ops.conv1d_cl(x, w, b, k, st, pd))
    same = torch.equal(o[0][1], o[1][1]); bad += not same
    tout = (T + 2 * pd - k) // st + 1
    fl = 2 * R * tout * cout * k * cin / 1e6
    print(f"conv R={R} T={T} cin={cin} cout={cout} k={k} s={st}: one-tile {o[0][0]:.0f} us ({fl / o[0][0] / PEAK:.3f}) -> persistent {o[1][0]:.0f} us "
          f"({fl / o[1][0] / PEAK:.3f} of peak); bit-identical {same}", flush=True)
    del x, w, b, o
# the whole exact-f32 forward (q/k/v = three problems per launch, residual out-projections, convs) at 64 samples of the headline shape
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
sd = synth.make_state_dict(cfg, 42)
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
m = m.cuda().eval(); m.precision = "f32"
B = 64
obj = torch.randn(B, 64, 32, 256, device="cuda"); lang = torch.randn(B, 16, 1024, device="cuda")
def fwd():
    with torch.no_grad():
        return m(obj, lang)
o = ab(fwd)
same = torch.equal(o[0][1][0], o[1][1][0]) and torch.equal(o[0][1][1], o[1][1][1]); bad += not same
print(f"forward f32, {B} samples: one-tile {o[0][0] / 1e3:.2f} ms -> persistent {o[1][0] / 1e3:.2f} ms ({B / o[1][0] * 1e6:.0f} samples/s); bit-identical {same}", flush=True)
# repeatability of the persistent kernel
ys = [fwd()[0].clone() for _ in range(5)]
rep = all(torch.equal(ys[0], y) for y in ys[1:]); bad += not rep
print("persistent forward repeatable over 5 calls:", rep)
print("FAILURES", bad)
sys.exit(1 if bad else 0)
