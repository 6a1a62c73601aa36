#!/bin/bash
# PMC passes over tools/attn_pmc_target.py (one counter group per pass; never combined with tracing).
# usage: tools/pmc_attn.sh <tag> <obj|motion|o2l> [B N Tp]      (SOLA_TUNE is passed through)
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/pmc_$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out" -o p$i -- python3 "$repo/tools/attn_pmc_target.py" "$@" > "$out/p$i.log" 2>&1
done
cd "$repo" && python3 tools/pmc_summary.py "$out" attn
