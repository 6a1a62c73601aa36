"""Per-run kernel durations from a rocprofv3 kernel trace: consecutive launches of one kernel (same name and grid) form a run.

    python tools/trace_runs.py gpurun_out/prof_<tag>/stats_kernel_trace.csv [substring of the kernel names to keep = attn]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
keep = sys.argv[2] if len(sys.argv) > 2 else "attn"
rows = [r for r in rows if keep in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
runs = []
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    key = (name, r["Grid_Size_X"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if runs and runs[-1][0] == key:
        runs[-1][1].append(d)
    else:
        runs.append([key, [d]])
for key, v in runs:
    s = sorted(v)
    print(f"{key[0][:60]:60s} grid {key[1]:>9s}  n {len(v):3d}  min {s[0]:7.1f}  median {s[len(s) // 2]:7.1f} us")
