#!/bin/bash
# rocprofv3 evidence for one bench.py configuration: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their
# own passes (never combined with the trace domains).  Usage, on the GPU box:
#   tools/profile_bench.sh <tag> [bench.py args...]      -> gpurun_out/prof_<tag>/{stats_*,pmc_fetch_*,pmc_write_*}.csv
# then `python profiles/summarize_rocprof.py gpurun_out/prof_<tag>` gives the per-kernel table kept under profiles/.
set -u
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# the headline leg only (split-f16 steps + the exact-f32 leg): the stress / ragged / IoU / 16-bit / training legs would mix other shapes
# into the per-kernel means
args="--steps 10 --warmup 2 --cpu-seconds 0 --extra-legs 0 --train-steps 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats -- python3 "$repo/bench.py" $args > "$out/bench_under_rocprof.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out" -o pmc_fetch -- python3 "$repo/bench.py" $args > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out" -o pmc_write -- python3 "$repo/bench.py" $args > "$out/pmc_write.log" 2>&1
cd "$repo" && python3 profiles/summarize_rocprof.py "$out" > "$out/summary.md"
tail -n 40 "$out/summary.md"
