"""Per-kernel time of the LAST training step in a rocprofv3 kernel trace; the number of kernels per step is found from the
distance between the last two launches of a once-per-step kernel (masked-MSE reduction).
  python tools/last_step_auto.py trace_kernel_trace.csv [top]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
marks = [i for i, r in enumerate(rows) if "mse_reduce_kernel" in r["Kernel_Name"]]
n = marks[-1] - marks[-2]
last = rows[marks[-2] + 1: marks[-1] + 1]
# rotate so that the window is one whole step (the marker sits mid-step; any window of n consecutive kernels is one step's worth)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:100]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
print(f"one step = {n} kernels: busy {tot / 1e3:.2f} ms, span {span / 1e3:.2f} ms")
fam = collections.defaultdict(lambda: [0, 0.0])
for k, (c, t) in agg.items():
    f = ("gemm" if k.startswith("gemm_kernel") else "attention" if k.startswith("attn") else "lora_grad" if "lora_grad" in k else
         "groupnorm" if k.startswith("gn_") else "layernorm" if k.startswith("ln_") else "geglu" if k.startswith("geglu") else
         "torch" if ("at::" in k or "rocclr" in k) else "other")
    fam[f][0] += c
    fam[f][1] += t
print("families:", ", ".join(f"{f} {c} launches {t / 1e3:.2f} ms" for f, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k:100s} {c:5d} {t / 1e3:7.3f} ms  avg {t / c:7.1f} us")
