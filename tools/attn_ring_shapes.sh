#!/bin/bash
# attn_ring.hip vs attn_simple.hip per shape class, kernel times from rocprofv3 traces (GPU box): tools/attn_ring_shapes.sh
for dims in 256,64,4 256,80,4 32,128,16 512,16,4 128,100,8 64,44,14; do
  export RING_DIMS=$dims
  tools/prof_stats.sh rs_$dims tools/attn_ring_diag.py attn_ring=0 attn_ring=1 attn_ring=0 attn_ring=1 > /dev/null 2>&1
  echo "== B,N,T' = $dims"
  python tools/trace_runs.py gpurun_out/prof_rs_$dims/stats_kernel_trace.csv
done
