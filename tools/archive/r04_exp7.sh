#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_ragged_train.py tests/test_gpu_forward.py tests/test_gpu_kernels.py tests/test_gpu_edge.py -x -q -m gpu > $O/tests_fused.txt 2>&1; tail -4 $O/tests_fused.txt
for r in 0 4096; do SOLA_TUNE=bwd_side_rows=$r timeout 200 python tools/train_one_probe.py 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 200 python tools/archive/graph_probe.py 2>&1 | grep -v amdgpu.ids | tail -6
