"""Largest idle gaps between consecutive kernels of the LAST step of a rocprofv3 kernel trace (who ends, how long nothing runs, who starts).
  python tools/step_gaps.py trace_kernel_trace.csv [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 15
marks = [i for i, r in enumerate(rows) if "mse_reduce_kernel" in r["Kernel_Name"]]
step = rows[marks[-2] + 1: marks[-1] + 1]
nm = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]  # noqa: E731
gaps = []
end = int(step[0]["End_Timestamp"])
for i in range(1, len(step)):
    st = int(step[i]["Start_Timestamp"])
    gaps.append(((st - end) / 1e3, i))
    end = max(end, int(step[i]["End_Timestamp"]))
tot = sum(g for g, _ in gaps if g > 0)
print(f"{len(step)} kernels, idle between kernels {tot / 1e3:.3f} ms; gaps > 3 us: {sum(1 for g, _ in gaps if g > 3)} summing {sum(g for g, _ in gaps if g > 3) / 1e3:.3f} ms; median gap {sorted(g for g, _ in gaps)[len(gaps) // 2]:.2f} us")
for g, i in sorted(gaps, reverse=True)[:top]:
    print(f"{g:9.1f} us   #{i:5d}  after {nm(step[i - 1])}   before {nm(step[i])}")
