"""bench.py under two builds of the library, interleaved (kernel experiments).  usage: bench_lib_ab.py libA.so libB.so [bench args]"""
import json, os, subprocess, sys
la, lb = sys.argv[1], sys.argv[2]
res = {la: [], lb: []}
for rnd in range(3):
    for lib in (la, lb):
        env = dict(os.environ, SOLA_HIP_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, "bench.py", "--cpu-seconds", "0", "--steps", "20", "--warmup", "3"] + sys.argv[3:], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line: print(p.stdout[-1500:], p.stderr[-1500:]); sys.exit(1)
        j = json.loads(line[-1]); res[lib].append((j["ms_per_step"], j["roofline"]["avg_launch_us"]))
for lib in (la, lb): print(lib, " ".join(f"{m:.2f}ms/{u:.0f}us" for m, u in res[lib]), " best", min(m for m, _ in res[lib]))
