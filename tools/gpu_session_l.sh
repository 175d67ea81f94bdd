#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/sessL
mkdir -p $O
run() { env "$@" timeout 400 python $R/bench.py $EXTRA --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu --steps 30 --warmup 5 2>$O/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$* $EXTRA', d['ms_per_step'], d['config'].get('final_loss'))"; }
{
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_gemmtouch.so
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_gemmtouch.so
EXTRA="--config sd15"
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_gemmtouch.so
EXTRA="--full-ft"
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_gemmtouch.so
} 2>&1 | tee $O/step_ab.txt
