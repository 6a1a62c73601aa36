#!/bin/bash
repo=${GRAFT_REPO_ROOT:-/root/repo}
make -C "$repo/sola_amd/csrc" EXPERIMENTS=1 -j32 > "$repo/gpurun_out/exp_build.log" 2>&1 || { tail -5 "$repo/gpurun_out/exp_build.log"; exit 1; }
for rep in 1 2 3; do for ord in 0 1; do
python3 "$repo/bench.py" --extra-legs 0 --train-steps 0 --cpu-seconds 0 --steps 30 --warmup 5 --tune gemm_order=$ord 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gemm_order', $ord, 'ms/step', o['ms_per_step'], 'value', o['value'], 'gemm_us', o['roofline'].get('avg_launch_us'), 'gemm ms', o['kernel_ms_per_step'].get('gemm_split256'))
"
done; done
