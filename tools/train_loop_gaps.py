"""What runs between two graph replays of the train() loop: kernels and idle time from the first kernel after a step's last graph node
(the AdamW / shadow refresh tail) to the first node of the next replay, averaged over the steady-state steps of a rocprofv3 kernel trace.
  python tools/train_loop_gaps.py trace_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]  # noqa: E731
marks = [i for i, r in enumerate(rows) if "mse_reduce_kernel" in r["Kernel_Name"]]
# a step's graph begins at the embed_gather / add_noise kernels; take "first sdlt kernel after a run of torch kernels" as the replay start
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a + 1: b + 1]
    # last kernel of the previous replay = last non-torch kernel before a run of at:: kernels that precedes this segment's body
    idx = [k for k, r in enumerate(seg) if "at::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
    steps.append((seg, idx))
span = [int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s, _ in steps]
print(f"{len(steps)} steps, mean mse->mse period {sum(int(b[0][-1]['End_Timestamp']) - int(a[0][-1]['End_Timestamp']) for a, b in zip(steps[:-1], steps[1:])) / 1e6 / (len(steps) - 1):.3f} ms")
seg, idx = steps[len(steps) // 2]
tor = collections.Counter()
tt = 0
for k in idx:
    tor[nm(seg[k])] += 1
    tt += int(seg[k]["End_Timestamp"]) - int(seg[k]["Start_Timestamp"])
print(f"torch / copy kernels inside one period: {len(idx)}, busy {tt / 1e3:.1f} us")
for k, v in tor.most_common(12):
    print(f"   {v:3d}  {k}")
gaps = []
end = int(seg[0]["End_Timestamp"])
for i in range(1, len(seg)):
    st = int(seg[i]["Start_Timestamp"])
    if st - end > 3000:
        gaps.append(((st - end) / 1e3, nm(seg[i - 1]), nm(seg[i])))
    end = max(end, int(seg[i]["End_Timestamp"]))
print(f"idle gaps > 3 us in that period: {len(gaps)} summing {sum(g for g, _, _ in gaps) / 1e3:.3f} ms")
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"{g:9.1f} us  after {a[:50]}  before {b[:50]}")
