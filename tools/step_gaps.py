"""Idle holes inside the LAST training step of a rocprofv3 kernel trace: every gap of more than 15 us between the end of everything launched so far and the next
kernel's start, with its offset from the step's first kernel and its position in the launch order.  python tools/step_gaps.py trace_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "mse_reduce_kernel" in r["Kernel_Name"]]
n = marks[-1] - marks[-2]
for back in (2, 1):          # the last two whole steps (marker to marker)
    seg = rows[marks[-back - 1]: marks[-back]]
    t0, hi = int(seg[0]["Start_Timestamp"]), int(seg[0]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
    print(f"step -{back}: {len(seg)} launches, span {(int(seg[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms, busy {busy:.3f} ms")
    for i, (a, b) in enumerate(zip(seg, seg[1:])):
        hi = max(hi, int(a["End_Timestamp"]))
        g = (int(b["Start_Timestamp"]) - hi) / 1e3
        if g > 15:
            nm = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:50]  # noqa: E731
            print(f"   hole {g:7.1f} us at +{(hi - t0) / 1e3:9.1f} us, after launch {i} ({nm(a)}) before ({nm(b)})")
