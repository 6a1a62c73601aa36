#!/bin/bash
# round-4 closing batch: the whole GPU suite, the default bench line, a two-rank (gloo, one GPU) bench line
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
cp gpurun_out/bench_full.json $O/bench_full.json
SOLA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --batch 64 --train-steps 2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; tail -c 1500 $O/bench_gloo2.json; echo; tail -3 $O/bench_gloo2.err
