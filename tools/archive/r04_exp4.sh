#!/bin/bash
# round-4 batch 4: headline step under the tile-order / loader-wave switches (SOLA_TUNE), interleaved twice
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
O=gpurun_out/r04
for rnd in 1 2; do
 for t in "" "gemm_order=1" "gemm_order=1,gemm_ld=1"; do
  SOLA_TUNE=$t timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --extra-legs 0 --train-steps 0 > $O/bench_tune.json 2> $O/bench_tune.err
  python - "$t" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04/bench_tune.json").read().strip().split("\n")[-1])
print(f"SOLA_TUNE={sys.argv[1]!r}: {d['value']:.0f} samples/s  {d['ms_per_step']:.3f} ms/step  frac_executed {d['roofline']['frac_executed']:.4f}", flush=True)
PY
 done
done
