#!/bin/bash
# PMC passes over the ragged bf16 training step (tools/train_ragged_probe.py 64 bf16), one counter group per pass, never combined with tracing;
# the summary lists the attention kernels.  usage (GPU box): tools/pmc_train_attn.sh <tag> [tune key=value,...]
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/pmc_$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out" -o p$i -- python3 "$repo/tools/train_ragged_probe.py" 64 bf16 "$@" > "$out/p$i.log" 2>&1
done
cd "$repo" && python3 tools/pmc_summary.py "$out" attn
