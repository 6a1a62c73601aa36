#!/bin/bash
# 16-bit storage mode, attention sites at C4 (32 x 128 x 128) and the headline shape: kernel times per sola_tune setting (GPU box)
#   tools/f16_attn_ab.sh "attn_f16_qpb=1" "attn_f16_qpb=4" ...
for t in "$@"; do
  export SOLA_TUNE=$t
  for cfg in "32 128 128" "256 64 32"; do
    tools/prof_stats.sh f16ab tools/f16_target.py $cfg 30 > /dev/null 2>&1
    echo "== $t | B N T = $cfg"
    python tools/trace_runs.py gpurun_out/prof_f16ab/stats_kernel_trace.csv > /tmp/tr_f16ab.txt; tail -3 /tmp/tr_f16ab.txt
  done
done
