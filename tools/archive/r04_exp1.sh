#!/bin/bash
# round-4 experiment batch 1: in-kernel trace of the persistent split GEMM, CU start stagger, W-affine tile order
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
O=gpurun_out/r04
python tools/gemm_trace.py > $O/trace_default.txt 2>&1
python tools/gemm_trace.py 65536x1024x1024 gemm_ablate=4 > $O/trace_noepi.txt 2>&1
python tools/gemm_trace.py 65536x1024x1024 gemm_ablate=1 > $O/trace_nodma.txt 2>&1
python tools/gemm_trace.py 65536x1024x1024 gemm_stagger=40 > $O/trace_stagger40.txt 2>&1
for s in 20 40 80; do
  SHAPES=65536x1024x1024x0x0,65536x1024x1024x1x0,262144x512x768x0x0 python tools/gemm_ab_probe.py gemm_stagger 0 $s > $O/ab_stagger$s.txt 2>&1
done
SHAPES=65536x3072x1024x0x0,65536x2048x1024x0x0 python tools/gemm_ab_probe.py gemm_order 0 1 > $O/ab_order.txt 2>&1
python bench.py > $O/bench_base.json 2> $O/bench_base.err
tail -n 30 $O/trace_default.txt $O/ab_*.txt
