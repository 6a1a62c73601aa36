#!/bin/bash
# rocprofv3 --kernel-trace --stats of one python script (per-kernel time table); usage on the GPU box:
#   tools/prof_stats.sh <tag> <script.py> [args...]   -> gpurun_out/prof_<tag>/stats_kernel_stats.csv + a top-40 table on stdout
set -u
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/prof_$tag
mkdir -p "$out"
script=$repo/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats -- python3 "$script" "$@" > "$out/run.log" 2>&1
tail -n 8 "$out/run.log"
python3 - "$out/stats_kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:45]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)[:110]
    print(f'{float(r["TotalDurationNs"])/1e6:9.3f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:9.1f} us avg {100*float(r["TotalDurationNs"])/tot:5.1f}%  {n}')
PY
