#!/bin/bash
# The BASELINE.json shapes other than the headline one, each at a batch that fills the GPU: one bench.py JSON line per shape
# (gpurun_out/r01_shapes.jsonl).  Columns: B N T.
out=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/r01_shapes.jsonl; : > "$out"
while read -r B N T; do
  python bench.py --batch "$B" --tracks "$N" --frames "$T" --steps 10 --cpu-seconds 0 2>/dev/null | tail -1 >> "$out"
done <<'CFG'
4096 8 8
1024 16 32
192 80 32
256 64 32
32 128 128
CFG
python - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line); c = d["config"]
    print(f"B={c['batch_per_gpu']:5d} N={c['tracks']:3d} T={c['frames']:3d}: {d['value']:9.1f} samples/s  {d['model_tflops']:6.1f} model TFLOP/s  "
          f"gemm frac {d['roofline']['frac']:.3f}  attn {d['roofline_attention']['achieved']:.0f} GB/s ({d['roofline_attention']['frac']:.3f})  exact-f32 {d['exact_f32_mode']['value']:.1f}")
PY
