"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run: share, calls, average duration.  usage: top_kernels.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{float(r['TotalDurationNs']) / tot * 100:5.1f}% calls {r['Calls']:>6} avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:120]}")
