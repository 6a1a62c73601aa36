#!/bin/bash
# effective clock + MFMA duty of the split GEMM (variant $1) under ablation $2: GRBM_GUI_ACTIVE / kernel duration
repo=${GRAFT_REPO_ROOT:-/root/repo}; out=$repo/gpurun_out/pmc_clock_$1_$2; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export SOLA_ABLATE=$2
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d "$out" -o p -- python3 "$repo/tools/gemm_pmc_target.py" $1 > "$out/p.log" 2>&1
python3 - "$out/p_counter_collection.csv" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_nt_split" in r["Kernel_Name"]:
        d[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in d.items():
    val = sum(x for x, _ in v) / len(v); ns = sum(t for _, t in v) / len(v)
    print(k, round(val), "dur_us", round(ns / 1e3, 1), "per-XCD cycles/us", round(val / 8 / (ns / 1e3), 1))
PY
