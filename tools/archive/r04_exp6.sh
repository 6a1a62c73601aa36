#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_ragged_train.py tests/test_gpu_dist.py -x -q -m gpu > $O/tests_side.txt 2>&1; tail -4 $O/tests_side.txt
for r in 0 4096; do SOLA_TUNE=bwd_side_rows=$r timeout 200 python tools/train_one_probe.py 2>&1 | grep -v amdgpu.ids | tail -2; done
