#!/usr/bin/env python3
"""Summarise a rocprofv3 run directory (kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes) per kernel.

    python profiles/summarize_rocprof.py gpurun_out/prof_r01 > profiles/r01_summary.md

FETCH_SIZE / WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 64 B per
128-B request for wide coalesced reads, so the read side is doubled ("fetch x2" column); WRITE_SIZE is uncalibrated
and shown raw.  Everything is per launch (mean over the launches of that kernel in the pass).
"""
import collections
import csv
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def pmc(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    d = sys.argv[1]
    stats = list(csv.DictReader(open(os.path.join(d, "stats_kernel_stats.csv"))))
    fetch = pmc(os.path.join(d, "pmc_fetch_counter_collection.csv"), "FETCH_SIZE")
    write = pmc(os.path.join(d, "pmc_write_counter_collection.csv"), "WRITE_SIZE")
    print("| kernel | calls | avg us | % time | FETCH_SIZE MiB/launch (raw) | fetch x2 MiB | WRITE_SIZE MiB/launch |")
    print("|---|---|---|---|---|---|---|")
    for r in stats:
        k = short(r["Name"])
        if k.startswith("at::") or "rocclr" in k:
            continue
        f = fetch.get(k)
        w = write.get(k)
        fm = f[1] / f[0] / 1024 if f and f[0] else float("nan")
        wm = w[1] / w[0] / 1024 if w and w[0] else float("nan")
        print(f"| {k} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} | {fm:.2f} | {2 * fm:.2f} | {wm:.2f} |")


if __name__ == "__main__":
    main()
