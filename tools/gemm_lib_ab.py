"""A/B of two builds of libsola_hip.so (kernel experiments): runs tools/gemm_ab_probe-like timings in two child processes
per round, interleaved.  usage: gemm_lib_ab.py libA.so libB.so [MxNxKxRESxOSP,...]"""
import os, subprocess, sys, json
child = r'''
import sys, json, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
out = {}
for (M, N, K, res, osp) in shapes:
    a = ops.cast_sp16(torch.randn(M, K, device="cuda")); w = ops.cast_sp16(torch.randn(N, K, device="cuda") * 0.03, 64.0)
    b = torch.randn(N, device="cuda"); r = ops.cast_sp16(torch.randn(M, N, device="cuda")) if res else None
    fn = lambda: ops.gemm_nt_split(a, w, b, r, True, 1 / 64, bool(osp))
    ts = []
    for rnd in range(5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    out[f"{M}x{N}x{K} res={res} osp={osp}"] = ts
    del a, w, r
print("RESULT " + json.dumps(out))
'''
la, lb = sys.argv[1], sys.argv[2]
shapes = sys.argv[3] if len(sys.argv) > 3 else "65536x1024x1024x0x0,65536x1024x1024x1x0,65536x1024x3072x0x0,262144x512x768x0x0,65536x1024x512x0x1"
res = {la: {}, lb: {}}
for rnd in range(3):
    for lib in (la, lb):
        env = dict(os.environ, SOLA_HIP_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, "-c", child, shapes], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not line: print(p.stdout[-2000:], p.stderr[-2000:]); sys.exit(1)
        for k, v in json.loads(line[0][7:]).items(): res[lib].setdefault(k, []).extend(v)
for k in res[la]:
    ma, mb = min(res[la][k]), min(res[lb][k])
    print(f"{k}: A min {ma:.1f} us  B min {mb:.1f} us  B/A {mb/ma:.3f}")
