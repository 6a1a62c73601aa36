#!/bin/bash
# PMC passes over the headline leg of bench.py (one counter group per pass; never combined with tracing): instruction mix and busy
# cycles per kernel, e.g. the fused conv + GroupNorm launches against the plain conv launches.
# usage: tools/pmc_bench.sh <tag> [bench.py args...]   -> gpurun_out/pmc_<tag>/p*_counter_collection.csv ; tools/pmc_summary.py <dir> <pattern>
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/pmc_$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out" -o p$i -- python3 "$repo/bench.py" --steps 4 --warmup 1 --cpu-seconds 0 --extra-legs 0 --train-steps 0 "$@" > "$out/p$i.log" 2>&1
done
cd "$repo" && python3 tools/pmc_summary.py "$out" "gemm_nt_split_glds_persist"
