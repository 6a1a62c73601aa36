#!/bin/bash
# round-4 closing batch (after the code-object audit, the few-row GEMM and the grouped one-sample backward)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; wc -c $O/bench_default.json
cp gpurun_out/bench_full.json $O/bench_full.json
SOLA_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --batch 64 --train-steps 2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; wc -c $O/bench_gloo2.json; tail -2 $O/bench_gloo2.err
python tools/infer_one_probe.py > $O/infer_one.txt 2>&1; cat $O/infer_one.txt
python tools/train_one_probe.py 2>&1 | tail -1 > $O/train_one.txt; cat $O/train_one.txt
