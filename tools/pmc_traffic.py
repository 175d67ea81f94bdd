"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of `bench.py` into
HBM-side bytes per training step, per kernel family.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
counts 128-B requests at 64 B -> doubled; WRITE_SIZE is uncalibrated and taken as reported.  Units of both counters: KiB.
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv, sys, json, collections, re
def per_step(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "mse_reduce_kernel" in r["Kernel_Name"]]
    step = rows[marks[-2]:marks[-1]]
    fam = collections.defaultdict(float)
    for r in step:
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
        f = "gemm" if k.startswith("gemm_kernel") else "attn" if k.startswith("attn") else "lora_grad" if "lora_grad" in k else \
            "norm" if k.startswith(("gn_", "ln_")) else "torch" if "at::" in k else "other"
        fam[f] += float(r["Counter_Value"]) * 1024
    return fam, len(step)
fetch, n = per_step(sys.argv[1])
write, _ = per_step(sys.argv[2])
out = {"kernels_per_step": n,
       "fetch_bytes_per_step": 2 * sum(fetch.values()), "write_bytes_per_step": sum(write.values()),
       "fetch_bytes_by_family": {k: 2 * v for k, v in sorted(fetch.items(), key=lambda kv: -kv[1])},
       "write_bytes_by_family": {k: v for k, v in sorted(write.items(), key=lambda kv: -kv[1])},
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --no-graph --no-cpu-baseline --steps 2 --warmup 1` (eager launches of the same kernels; counter collection over the whole-step hipGraph does not finish); "
               "last full step; FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE as reported (uncalibrated); "
               "Infinity-Cache hits are included in both"}
out["traffic_bytes_per_step"] = out["fetch_bytes_per_step"] + out["write_bytes_per_step"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
