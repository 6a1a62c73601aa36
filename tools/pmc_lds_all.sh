#!/bin/bash
# LDS bank-conflict share per kernel over one bench run (PMC pass, no tracing)
repo=${GRAFT_REPO_ROOT:-/root/repo}; out=$repo/gpurun_out/pmc_lds_all; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d "$out" -o p -- python3 "$repo/bench.py" --batch 64 --steps 3 --warmup 1 --cpu-seconds 0 > "$out/p.log" 2>&1
python3 - "$out/p_counter_collection.csv" <<'PY'
import csv, sys, collections, re
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"((?:gemm|attn|group_norm|score|loss|ws_|cast|lang)\w*(<[^>]*>)?)", r["Kernel_Name"])
    if not m: continue
    d[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); n[m.group(1)] += 1
for k, v in sorted(d.items()):
    act = v.get("SQ_LDS_IDX_ACTIVE", 0)
    print(f"{k:50s} launches {n[k] // 3:4d}  conflict/active = {v.get('SQ_LDS_BANK_CONFLICT', 0) / act if act else 0:.2f}  (active {act:.0f}, LDS insts {v.get('SQ_INSTS_LDS', 0):.0f})")
PY
