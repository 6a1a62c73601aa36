#!/bin/bash
# A/B of two builds of libsola_hip.so on the attention sites (GPU box): tools/attn_ab_libs.sh <libA.so> <libB.so>
# inter-object and object->language shapes at the headline dims, kernel times from rocprofv3 traces; every configuration runs twice
# (the first run of a process carries the clock ramp)
A=$1; B=$2
for shape in obj o2l; do
  export RING_SHAPE=$shape
  for lib in $A $B; do
    export SOLA_HIP_LIB=$PWD/$lib
    tag=ab_${shape}_$(basename $lib .so)
    tools/prof_stats.sh $tag tools/attn_ring_diag.py attn_simple_db=1 attn_simple_db=1 attn_simple_db=1 > /dev/null 2>&1
    echo "== $shape $lib"
    python tools/trace_runs.py gpurun_out/prof_$tag/stats_kernel_trace.csv
  done
done
