#!/bin/bash
# round-4 batch 3: grouped fragment schedule (spill-free persistent GEMM) - correctness and A/B against the build before it
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fast.py tests/test_gpu_kernels.py -x -q -m gpu > $O/tests_gemm.txt 2>&1; tail -3 $O/tests_gemm.txt
timeout 600 python tools/gemm_lib_ab.py build/libsola_r04_base.so sola_amd/lib/libsola_hip.so > $O/lib_ab_grouped.txt 2>&1; cat $O/lib_ab_grouped.txt
timeout 300 python bench.py > $O/bench_grouped.json 2> $O/bench_grouped.err
python - <<'PY'
import json
for f in ("bench_base", "bench_grouped"):
    try:
        d = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac_executed"])
    except Exception as e: print(f, "failed", e)
PY
