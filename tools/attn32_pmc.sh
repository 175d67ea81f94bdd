#!/bin/bash
# PMC counters of the self-attention kernels at one shape (two passes; gpurun refuses --pmc together with the trace domains other than --kernel-trace)
# usage: tools/attn32_pmc.sh N H [outfile]
R=$GRAFT_REPO_ROOT
out=${3:-$R/gpurun_out/attn32_pmc.txt}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ap1 /tmp/ap2 /tmp/ap3
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/ap1 -- python $R/tools/attn32_pmc.py $1 $2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d /tmp/ap2 -- python $R/tools/attn32_pmc.py $1 $2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --kernel-trace --output-format csv -d /tmp/ap3 -- python $R/tools/attn32_pmc.py $1 $2 > /dev/null 2>&1
python - >> $out <<PY
import csv, glob, collections
print("== N $1 H $2")
for d in ("/tmp/ap1", "/tmp/ap2", "/tmp/ap3"):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f:
        print("no counters in", d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "attn" in k:
            agg[k.replace("void (anonymous namespace)::", "")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(k, {n: round(sum(v[2:]) / max(1, len(v[2:]))) for n, v in c.items()})
PY
cat $out
