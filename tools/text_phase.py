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


report("text forward", step[:an])
if egr:
    # walk back from the last embed_grad while the kernels are text-encoder kernels (strip / ln_bwd / attention of 77 tokens / small torch ops)
    end = egr[-1]
    j = end
    while j > an and not any(t in step[j]["Kernel_Name"] for t in ("attn_splitsum_batch", "gn_bwd", "sum2x2", "lora_grad")):
        j -= 1
    report("text backward", step[j + 1: end + 1])
print(f"whole step: {n} launches, span {(int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6:.3f} ms")
