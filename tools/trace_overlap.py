"""From a rocprofv3 kernel trace of bench.py: take the last graph replay (delimited by the adamw kernels), report wall time,
summed kernel time, per-queue kernel counts and the time during which >1 kernel was resident."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "mse_reduce_kernel" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, multi, idle = 0, t0, 0, 0
for t, d in ev:
    if depth > 1: multi += t - last
    if depth == 0: idle += t - last
    depth += d; last = t
q = collections.Counter(r["Queue_Id"] for r in step)
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, kernels {len(step)}, summed kernel time {busy / 1e6:.2f} ms, >1 resident {multi / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms, queues {dict(q)}")
