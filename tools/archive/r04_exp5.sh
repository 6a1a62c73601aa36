#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_ragged_train.py tests/test_gpu_backward.py tests/test_gpu_f16.py tests/test_gpu_range.py -x -q -m gpu -s > $O/tests_parity.txt 2>&1; tail -5 $O/tests_parity.txt
grep -h "worst logit\|training forward\|operand training\|tokens worst" $O/tests_parity.txt | head -40
