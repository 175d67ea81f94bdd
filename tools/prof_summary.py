"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel family time per step."""
import csv, collections, sys, re
path, nsteps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r['Kernel_Name']
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[n][0] += d; agg[n][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"total kernel time {tot/1e6/nsteps:.2f} ms/step over {nsteps:g} steps")
fam = collections.defaultdict(float)
for k, v in agg.items():
    f = 'gemm' if k.startswith('gemm_kernel') else ('attn' if k.startswith('attn') else ('lora_grad' if 'lora_grad' in k else ('norm' if k.startswith(('gn_', 'ln_')) else ('torch' if 'at::' in k or 'rocclr' in k else 'other'))))
    fam[f] += v[0]
print({k: round(v/1e6/nsteps, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])})
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"{k[:70]:70s} calls/step {v[1]/nsteps:7.1f}  ms/step {v[0]/1e6/nsteps:7.3f}  avg {v[0]/v[1]/1e3:8.1f} us")
