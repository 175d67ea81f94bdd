#!/bin/bash
# round 5 session B: 2-D rotation (bit 4) and prologue touches (bit 5) of the packed wave-split-K kernels - probe + whole step; determinism probe
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/sessB
mkdir -p $O
for st in 1 17 33 49; do SDLT_WSK_STAGGER=$st timeout 300 python tools/wsk_pack_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee $O/probe.txt
run() { env "$@" timeout 400 python $R/bench.py --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu --steps 30 --warmup 5 2>$O/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['config'].get('final_loss'))"; }
{
run SDLT_WSK_STAGGER=1
run SDLT_WSK_STAGGER=17
run SDLT_WSK_STAGGER=33
run SDLT_WSK_STAGGER=49
run SDLT_WSK_STAGGER=1
run SDLT_WSK_STAGGER=49
} 2>&1 | tee $O/step_ab.txt
timeout 900 python tools/determinism_probe.py sdxl 128 1 > $O/determinism.txt 2>&1; tail -40 $O/determinism.txt
