"""One JSON + one per-kernel CSV for the LAST training step of a profiled `bench.py` run (rocprofv3):
  python tools/step_profile.py <kernel_trace.csv> <fetch counter_collection.csv | -> <write counter_collection.csv | -> <out.json> <out_per_kernel.csv> [commit] [command]
* kernel trace (graph replays, `--kernel-trace`): per-family and per-kernel time of the last step;
* PMC passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate runs of `bench.py --no-graph`: counter collection over the whole-step hipGraph
  does not finish): HBM-side bytes per kernel; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B requests at
  64 B -> doubled, WRITE_SIZE as reported (uncalibrated), unit KiB; Infinity-Cache hits are included in both.
`bench.py --profile-json <out.json>` turns the family times into GB/s against the algorithmic bytes (topology.hbm_bytes)."""
import collections
import os
import csv
import json
import re
import sys


def clean(k):
    return re.sub(r"\(anonymous namespace\)::", "", k).replace("void ", "")


def family(k):
    if k.startswith(("gemm_kernel", "wsk_kernel")):
        return "gemm"
    if k.startswith("strip_"):        # strip_kernel and strip_pair_kernel
        return "text_gemm"
    if k.startswith("attn"):
        return "attention"
    if "lora_grad" in k:
        return "lora_grad"
    if k.startswith("gn_"):
        return "groupnorm"
    if k.startswith("ln_"):
        return "layernorm"
    if k.startswith("geglu"):
        return "geglu"
    if k.startswith(("adamw_kernel", "shadow_kernel")):
        return "adamw"
    if "at::" in k or "rocclr" in k:
        return "torch"
    return "other"


def last_step(rows, key):
    rows.sort(key=lambda r: int(r[key]))
    marks = [i for i, r in enumerate(rows) if "mse_reduce_kernel" in r["Kernel_Name"]]
    return rows[marks[-2] + 1: marks[-1] + 1]


trace, fetch, write, out_json, out_csv = sys.argv[1:6]
commit = sys.argv[6] if len(sys.argv) > 6 else None
command = sys.argv[7] if len(sys.argv) > 7 else None
per = collections.defaultdict(lambda: dict(count=0, time_us=0.0, fetch_bytes=0.0, write_bytes=0.0))
step = last_step(list(csv.DictReader(open(trace))), "Start_Timestamp")
for r in step:
    k = clean(r["Kernel_Name"])
    per[k]["count"] += 1
    per[k]["time_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span_ms = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
n_pmc = None
for path, field, mult in ((fetch, "fetch_bytes", 2.0), (write, "write_bytes", 1.0)):
    if path == "-":
        continue
    st = last_step(list(csv.DictReader(open(path))), "Dispatch_Id")
    n_pmc = len(st)
    for r in st:
        per[clean(r["Kernel_Name"])][field] += float(r["Counter_Value"]) * 1024 * mult
fam_ms, fam_f, fam_w, fam_n = (collections.defaultdict(float) for _ in range(4))
for k, v in per.items():
    f = family(k)
    fam_ms[f] += v["time_us"] / 1e3
    fam_f[f] += v["fetch_bytes"]
    fam_w[f] += v["write_bytes"]
    fam_n[f] += v["count"]
srt = lambda d: {k: v for k, v in sorted(d.items(), key=lambda kv: -kv[1])}  # noqa: E731
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_profile_sha import kernel_sources_sha  # noqa: E402

out = {"commit": commit, "kernel_sources_sha16": kernel_sources_sha(), "command": command, "kernels_per_step": len(step), "kernels_per_step_pmc_pass": n_pmc, "step_span_ms": span_ms,
       "step_busy_ms": sum(fam_ms.values()), "family_ms": srt(fam_ms), "family_launches": srt(fam_n),
       "fetch_bytes_by_family": srt(fam_f), "write_bytes_by_family": srt(fam_w),
       "fetch_bytes_per_step": sum(fam_f.values()), "write_bytes_per_step": sum(fam_w.values()),
       "traffic_bytes_per_step": sum(fam_f.values()) + sum(fam_w.values()),
       "note": "times: rocprofv3 --kernel-trace of the graph-replayed step (last step); bytes: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over "
               "`bench.py --no-graph` (eager launches of the same kernels), FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B), WRITE_SIZE as reported "
               "(uncalibrated), Infinity-Cache hits included; per-kernel rows in the CSV beside this file"}
json.dump(out, open(out_json, "w"), indent=1)
with open(out_csv, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "family", "launches", "time_us", "fetch_bytes", "write_bytes"])
    for k, v in sorted(per.items(), key=lambda kv: -kv[1]["time_us"]):
        w.writerow([k, family(k), v["count"], round(v["time_us"], 2), int(v["fetch_bytes"]), int(v["write_bytes"])])
print(json.dumps(out, indent=1))
