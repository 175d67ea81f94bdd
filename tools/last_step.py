"""Per-kernel time of the LAST training step in a rocprofv3 kernel trace (excludes init / capture): python tools/last_step.py trace.csv n_kernels_per_step"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
last = rows[-n:]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    k = r["Kernel_Name"][:90]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
print(f"last {n} kernels: busy {tot / 1e3:.2f} ms, span {span / 1e3:.2f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k:90s} {c:5d} {t / 1e3:7.3f} ms  avg {t / c:7.1f} us")
