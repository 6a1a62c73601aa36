"""Mean per-launch value of every PMC counter of the gemm kernels in a tools/pmc_gemm.sh output directory."""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        pat = sys.argv[2] if len(sys.argv) > 2 else "gemm"
        m = re.search(r"(" + pat + r"_\w+(<[^>]*>)?)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:36s} {sum(v)/len(v):16.0f}  (n={len(v)})")
