#!/bin/bash
# Per GEMM signature of the step: HBM-side bytes fetched (rocprofv3 --pmc FETCH_SIZE, doubled: gfx950 tallies 128-B requests at 64 B) and written
# (WRITE_SIZE) per launch against the algorithmic operand bytes (every operand once) - the over-fetch column VERDICT r3 item 2c asks for.
# usage (on the GPU box): bash tools/gemm_fetch_ratio.sh [out.txt] [census timing file to join, optional]
R=$GRAFT_REPO_ROOT
out=${1:-$R/gpurun_out/r04_gemm_fetch_ratio.txt}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gf_f /tmp/gf_w /tmp/gf_order.json
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/gf_f -- python $R/tools/gemm_census.py --pmc-order /tmp/gf_order.json > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/gf_w -- python $R/tools/gemm_census.py --pmc-order /tmp/gf_order_w.json > /dev/null 2>&1
python - > $out <<PY
import csv, glob, json
order = json.load(open("/tmp/gf_order.json"))
def groups(d):
    f = glob.glob(d + "/*/*counter_collection.csv")
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "spin_kernel" in r["Kernel_Name"] or "sleep" in r["Kernel_Name"].lower()]
    marks = marks[-(len(order) + 1):]            # the PMC section is the LAST len(order)+1 markers of the run
    out = []
    for a, b in zip(marks, marks[1:]):
        g = [r for r in rows[a + 1:b] if "gemm_kernel" in r["Kernel_Name"] or "wsk_kernel" in r["Kernel_Name"]]
        out.append((sum(float(r["Counter_Value"]) for r in g) * 1024.0, len(g), sorted({r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:34] for r in g})))
    return out
gf, gw = groups("/tmp/gf_f"), groups("/tmp/gf_w")
print("# per launch; fetch = FETCH_SIZE x 2 (gfx950 correction), write = WRITE_SIZE as reported; algorithmic = every operand once (X, W, residual | the output)")
print(f"{'signature':52s} {'calls':>5s} {'fetch MB':>9s} {'alg MB':>8s} {'ratio':>6s} {'write MB':>9s} {'alg MB':>8s} {'ratio':>6s}  kernel")
tot_f = tot_af = 0.0
rows = []
for o, (fb, nf, kn), (wb, nw, _) in zip(order, gf, gw):
    if not nf:
        continue
    f, w = 2.0 * fb / nf, wb / max(nw, 1)
    rows.append((2.0 * fb, o, f, w, kn))
    tot_f += 2.0 * fb; tot_af += o["algorithmic_fetch_bytes"] * o["calls"]
for _, o, f, w, kn in sorted(rows, key=lambda r: -r[0])[:60]:
    print(f"{o['sig']:52s} {o['calls']:5d} {f / 1e6:9.2f} {o['algorithmic_fetch_bytes'] / 1e6:8.2f} {f / o['algorithmic_fetch_bytes']:6.2f} {w / 1e6:9.2f} {o['algorithmic_write_bytes'] / 1e6:8.2f} {w / o['algorithmic_write_bytes']:6.2f}  {' | '.join(kn)}")
print(f"# all GEMM launches of the step: fetched {tot_f / 1e9:.2f} GB, algorithmic {tot_af / 1e9:.2f} GB, ratio {tot_f / tot_af:.2f}")
PY
cat $out
