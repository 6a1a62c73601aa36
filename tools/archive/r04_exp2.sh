#!/bin/bash
# round-4 experiment batch 2: loader-wave DMA stream (gemm_ld 1 / 2)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 300 python tools/gemm_ld_check.py > $O/ld_check.txt 2>&1
S=65536x1024x1024x0x0,65536x1024x3072x0x0,262144x512x768x0x0,65536x3072x1024x0x0
SHAPES=$S timeout 300 python tools/gemm_ab_probe.py gemm_ld 0 1 > $O/ab_ld1.txt 2>&1
SHAPES=$S timeout 300 python tools/gemm_ab_probe.py gemm_ld 0 2 > $O/ab_ld2.txt 2>&1
timeout 300 python tools/gemm_trace.py 65536x1024x1024 gemm_ld=1 > $O/trace_ld1.txt 2>&1
timeout 300 python tools/gemm_trace.py 65536x1024x1024 gemm_ld=2 > $O/trace_ld2.txt 2>&1
cat $O/ld_check.txt $O/ab_ld1.txt $O/ab_ld2.txt; grep -v "^   tile" $O/trace_ld1.txt $O/trace_ld2.txt
