#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py at one shape (timed region only): tools/prof_bench_shape.sh <tag> <B> <N> <T>
set -u
tag=$1; B=$2; N=$3; T=$4
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats -- python3 "$repo/bench.py" --batch "$B" --tracks "$N" --frames "$T" --steps 10 --warmup 2 --cpu-seconds 0 --extra-legs 0 --train-steps 0 > "$out/run.log" 2>&1
grep '^{"metric"' "$out/run.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline_attention']['frac'], d['kernel_ms_per_step'])"
python3 - "$out/stats_kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)[:100]
    print(f'{float(r["TotalDurationNs"])/1e6:9.3f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:9.1f} us avg {100*float(r["TotalDurationNs"])/tot:5.1f}%  {n}')
PY
