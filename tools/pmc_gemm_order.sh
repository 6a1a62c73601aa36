#!/bin/bash
# Round 6: L2-miss bytes (FETCH_SIZE, its own pass, no tracing) of the split-f16 GEMM launches of the headline step under the default tile order
# and under sola_tune "gemm_order" 1 (an XCD's 32 CUs share 8 row panels x 4 column tiles per round) - EXPERIMENTS=1 build only.
# usage on the GPU box: tools/pmc_gemm_order.sh   -> gpurun_out/pmc_gemm_order.txt
repo=${GRAFT_REPO_ROOT:-/root/repo}
make -C "$repo/sola_amd/csrc" EXPERIMENTS=1 -j32 > "$repo/gpurun_out/exp_build.log" 2>&1 || { tail -5 "$repo/gpurun_out/exp_build.log"; exit 1; }
cd /tmp && export TMPDIR=/tmp
for ord in 0 1; do
  out=$repo/gpurun_out/pmc_order_$ord; mkdir -p "$out"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out" -o pmc_fetch -- python3 "$repo/bench.py" --steps 6 --warmup 2 --cpu-seconds 0 --extra-legs 0 --train-steps 0 --precision f16x3 --tune gemm_order=$ord > "$out/run.log" 2>&1
  tail -c 400 "$out/run.log" | grep -o '"value":[0-9.]*,"unit"[^,]*,[^,]*,[^,]*,[^,]*,"ms_per_step":[0-9.]*' | head -1
done
python3 - "$repo" <<'PY' | tee "$repo/gpurun_out/pmc_gemm_order.txt"
import collections, csv, re, sys
repo = sys.argv[1]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^>]*>)?)", n)
    return m.group(1) if m else n[:60]
res = {}
for ord_ in (0, 1):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"{repo}/gpurun_out/pmc_order_{ord_}/pmc_fetch_counter_collection.csv")):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    res[ord_] = agg
print("FETCH_SIZE per launch, MiB (x2: gfx950 counts 64 B per 128-B request), split-f16 GEMM instantiations of the headline step")
print(f"{'kernel':70s} {'launches':>8s} {'order 0':>10s} {'order 1':>10s} {'ratio':>6s}")
for k in sorted(res[0], key=lambda k: -res[0][k][1]):
    if "gemm_nt_split" not in k:
        continue
    a, b = res[0][k], res[1].get(k, [0, 0.0])
    fa = 2 * a[1] / max(a[0], 1) / 1024
    fb = 2 * b[1] / max(b[0], 1) / 1024
    print(f"{k:70s} {a[0]:8d} {fa:10.1f} {fb:10.1f} {fb / fa if fa else 0:6.3f}")
PY
