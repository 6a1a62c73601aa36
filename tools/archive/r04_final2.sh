#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; wc -c $O/bench_default.json
cp gpurun_out/bench_full.json $O/bench_full.json
