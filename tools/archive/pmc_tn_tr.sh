#!/bin/bash
# HBM traffic and matrix-pipe duty of the row-major weight-gradient kernel against the NT kernel on transposed copies, one shape:
# separate --pmc passes over tools/archive/gemm_tn_tr_once.py (never combined with tracing).  usage: tools/archive/pmc_tn_tr.sh <tag> [M N K]
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/pmc_tntr_$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out" -o p$i -- python3 "$repo/tools/archive/gemm_tn_tr_once.py" "$@" > "$out/p$i.log" 2>&1
done
cd "$repo" && python3 tools/pmc_summary.py "$out" "gemm_(tn_tr|nt_split_glds_persist)"
