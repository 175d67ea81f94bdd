"""Text-encoder share of the LAST training step in a rocprofv3 kernel trace: the forward phase runs from the first embed_gather to the
add_noise kernel (start of the UNet), the backward phase from the end of the UNet backward (the last kernel before the first strip /
LayerNorm-backward run that ends in embed_grad) to the second embed_grad.  Prints span, launch count and per-kernel totals of each.
  python tools/text_phase.py trace_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")  # noqa: E731
marks = [i for i, r in enumerate(rows) if "mse_reduce_kernel" in r["Kernel_Name"]]
n = marks[-1] - marks[-2]
# one whole step = from the last first-embed_gather before the last marker
eg = [i for i, r in enumerate(rows) if "embed_gather_kernel" in r["Kernel_Name"]]
starts = [i for i in eg if i < marks[-1]]
s0 = starts[-2] if len(starts) >= 2 and starts[-1] - starts[-2] < n // 2 else starts[-1]
step = rows[s0: s0 + n]
an = next(i for i, r in enumerate(step) if "add_noise" in r["Kernel_Name"])
egr = [i for i, r in enumerate(step) if "embed_grad_kernel" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3  # noqa: E731


def report(title, seg):
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in seg:
        k = name(r)[:90]
        agg[k][0] += 1
        agg[k][1] += dur(r)
    print(f"{title}: {len(seg)} launches, span {span / 1e3:.3f} ms, busy {sum(v[1] for v in agg.values()) / 1e3:.3f} ms")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {k:90s} {c:5d} {t / 1e3:7.3f} ms  avg {t / c:6.1f} us")
    # where the idle time of the phase sits: the gaps between the end of everything launched so far and the next start
    gaps, hi = [], int(seg[0]["End_Timestamp"])
    for a, b in zip(seg, seg[1:]):
        hi = max(hi, int(a["End_Timestamp"]))
        g = (int(b["Start_Timestamp"]) - hi) / 1e3
        if g > 0:
            gaps.append((g, name(a)[:40], name(b)[:40]))
    if gaps:      # the launches around the largest hole, with their durations
        gi = max(range(len(seg) - 1), key=lambda i: int(seg[i + 1]["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in seg[max(0, i - 8): i + 1]))
        t0 = int(seg[0]["Start_Timestamp"])
        for x in seg[max(0, gi - 6): gi + 4]:
            print(f"      @{(int(x['Start_Timestamp']) - t0) / 1e3:9.1f} us  {dur(x):7.1f} us  q{x.get('Queue_Id', '?')} s{x.get('Stream_Id', '?')}  grid {x.get('Grid_Size_X', x.get('Grid_Size', '?'))}  {name(x)[:70]}")
    tot = sum(g for g, _, _ in gaps)
    big = sorted(gaps, reverse=True)[:6]
    print(f"   idle {tot / 1e3:.3f} ms in {len(gaps)} gaps (median {sorted(g for g, _, _ in gaps)[len(gaps) // 2] if gaps else 0:.2f} us); largest: " +
          "; ".join(f"{g:.1f} us after {a} before {b}" for g, a, b in big))


report("text forward", step[:an])
if egr:
    # walk back from the last embed_grad while the kernels are text-encoder kernels (strip / ln_bwd / attention of 77 tokens / small torch ops)
    end = egr[-1]
    j = end
    while j > an and not any(t in step[j]["Kernel_Name"] for t in ("attn_splitsum_batch", "gn_bwd", "sum2x2", "lora_grad")):
        j -= 1
    report("text backward", step[j + 1: end + 1])
print(f"whole step: {n} launches, span {(int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6:.3f} ms")
