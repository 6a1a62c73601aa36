#!/usr/bin/env python3
"""profiles/<tag>_summary.md (profiles/summarize_rocprof.py output) -> profiles/r01_traffic.json: HBM bytes per launch
(FETCH_SIZE x2 + WRITE_SIZE) and rocprofv3 average duration per kernel, plus the name of the GEMM instantiation that
takes the largest share of the step.  bench.py reads it for the `traffic` field when its batch matches.
usage: make_traffic_json.py profiles/r02_summary.md 256 [profiles/r02_traffic.json]"""
import json, sys
rows, share = {}, {}
for line in open(sys.argv[1]):
    c = [x.strip() for x in line.strip().strip('|').split('|')]
    if len(c) == 7 and c[1].isdigit():
        rows[c[0]] = {"hbm_bytes_per_launch": int((float(c[5]) + float(c[6])) * (1 << 20)), "rocprof_avg_us": float(c[2]), "calls": int(c[1])}
        share[c[0]] = float(c[3])
gemms = [k for k in rows if k.startswith("gemm_nt_split")]  # the exact-f32 leg of the same run is not the headline
# bench.py's roofline line covers every 256x256 launch of a step (its own profiler category); those are the instantiations
# of the persistent kernel, so their launch-weighted mean is the per-launch traffic / duration that goes with it
# (the instantiations that also apply GroupNorm in the epilogue - template argument GNT = 16 / 8 / 4 - are a category of their own)
def _gnt(name):  # <CONV, RMODE, CSP, PURE, GNT, NW>: rows per GroupNorm instance of the epilogue, 0 = none
    args = name[name.index("<") + 1:name.rindex(">")].split(",")
    return int(args[4]) if len(args) > 4 else 0
fused = [k for k in rows if k.startswith("gemm_nt_split_glds_persist_kernel") and _gnt(k) != 0]
p256 = [k for k in rows if (k.startswith("gemm_nt_split_glds_persist_kernel") and k not in fused) or k.startswith("gemm_nt_split_glds_kernel<4")]
if fused:
    calls = sum(rows[k]["calls"] for k in fused)
    rows["gemm_split256_gn (conv + GroupNorm launches)"] = {
        "hbm_bytes_per_launch": int(sum(rows[k]["hbm_bytes_per_launch"] * rows[k]["calls"] for k in fused) / calls),
        "rocprof_avg_us": round(sum(rows[k]["rocprof_avg_us"] * rows[k]["calls"] for k in fused) / calls, 1), "calls": calls, "instantiations": fused}
if p256:
    calls = sum(rows[k]["calls"] for k in p256)
    rows["gemm_split256 (all 256x256 launches)"] = {
        "hbm_bytes_per_launch": int(sum(rows[k]["hbm_bytes_per_launch"] * rows[k]["calls"] for k in p256) / calls),
        "rocprof_avg_us": round(sum(rows[k]["rocprof_avg_us"] * rows[k]["calls"] for k in p256) / calls, 1), "calls": calls,
        "instantiations": p256}
out = {"source": f"{sys.argv[1]}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 10 --warmup 2 "
                 f"--cpu-seconds 0 --extra-legs 0 --train-steps 0` (batch {sys.argv[2]}); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B "
                 "request), WRITE_SIZE raw; bytes per launch, mean over the launches of the kernel",
       "batch": int(sys.argv[2]), "dominant_gemm": "gemm_split256 (all 256x256 launches)" if p256 else (max(gemms, key=lambda k: share[k]) if gemms else None), "kernels": rows}
json.dump(out, open(sys.argv[3] if len(sys.argv) > 3 else "profiles/r02_traffic.json", "w"), indent=1)
print(out["dominant_gemm"], rows.get(out["dominant_gemm"]))
