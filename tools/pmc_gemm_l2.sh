#!/bin/bash
# L1->L2 request counts of one split-f16 GEMM launch, default kernel against the "gemm_k16" experiment (64-byte row pieces): is a
# 64-byte DMA row served as half a 128-byte line twice?  One counter group per pass; never combined with tracing.
# usage: tools/pmc_gemm_l2.sh <tag> [M N K]
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export SOLA_NORES=1
for k16 in 0 1; do
  export SOLA_K16=$k16
  out=$repo/gpurun_out/pmc_${tag}_k16_$k16; mkdir -p "$out"
  i=0
  for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" \
             "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d "$out" -o p$i -- python3 "$repo/tools/gemm_pmc_target.py" 4 "$@" > "$out/p$i.log" 2>&1
  done
  (cd "$repo" && python3 tools/pmc_summary.py "$out") > "$repo/gpurun_out/pmc_${tag}_k16_$k16.txt" 2>&1
done
