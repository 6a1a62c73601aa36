#!/bin/bash
# PMC passes over tools/gemm_f32_pmc_target.py (the exact-f32 MFMA GEMM; one counter group per pass; never combined with tracing).
# usage: tools/pmc_gemm_f32.sh <tag> [M N K] [variant]
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/pmc_$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out" -o p$i -- python3 "$repo/tools/gemm_f32_pmc_target.py" "$@" > "$out/p$i.log" 2>&1
done
cd "$repo" && python3 tools/pmc_summary.py "$out"
