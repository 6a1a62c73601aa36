#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_tests.log 2>&1; tail -5 gpurun_out/r04/gpu_tests.log
