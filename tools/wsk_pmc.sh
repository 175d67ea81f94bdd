R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wp1 /tmp/wp2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/wp1 -- python $R/tools/wsk_pmc.py ${1:-5120} > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d /tmp/wp2 -- python $R/tools/wsk_pmc.py ${1:-5120} > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in ("/tmp/wp1", "/tmp/wp2"):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f:
        print("no counters in", d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "wsk_kernel" in k or "gemm_kernel" in k:
            agg["wsk" if "wsk_kernel" in k else "tiled"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(k, {n: round(sum(v[4:]) / len(v[4:])) for n, v in c.items()})
PY
