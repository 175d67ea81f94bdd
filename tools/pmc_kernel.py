"""Per-kernel sums of the counters of ONE rocprofv3 PMC pass over the last full training step (bench.py --no-graph).
usage: pmc_kernel.py counter_collection.csv"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ids = sorted({int(r["Dispatch_Id"]) for r in rows if "mse_reduce_kernel" in r["Kernel_Name"]})
lo, hi = ids[-2], ids[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if lo <= int(r["Dispatch_Id"]) < hi:
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for v in agg.values() for c in v})
print("kernel".ljust(60), *[n[-22:].rjust(24) for n in names])
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:25]:
    print(k.ljust(60), *[f"{v.get(n, 0):24.4g}" for n in names])
