"""Per-kernel-family sums of arbitrary rocprofv3 PMC counters over the last full training step of `bench.py --no-graph`
(one counter_collection.csv per pass; a pass may hold several counters).
usage: pmc_family.py out.json pass1.csv [pass2.csv ...]"""
import collections, csv, json, re, sys


def fam_of(name):
    k = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    return "gemm" if k.startswith(("gemm_kernel", "wsk_kernel")) else "text_gemm" if k.startswith("strip_") else "attn" if k.startswith("attn") else "lora_grad" if "lora_grad" in k else \
        "norm" if k.startswith(("gn_", "ln_")) else "torch" if "at::" in k else "other"


out = collections.defaultdict(lambda: collections.defaultdict(float))
for path in sys.argv[2:]:
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if "mse_reduce_kernel" in r["Kernel_Name"]})
    lo, hi = ids[-2], ids[-1]
    for r in rows:
        if lo <= int(r["Dispatch_Id"]) < hi:
            out[r["Counter_Name"]][fam_of(r["Kernel_Name"])] += float(r["Counter_Value"])
res = {c: dict(sorted(f.items(), key=lambda kv: -kv[1])) for c, f in out.items()}
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps(res, indent=1))
