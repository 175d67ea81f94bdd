#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/sessH
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "norm or attention or attn" > $O/tests_norm.log 2>&1; tail -4 $O/tests_norm.log
run() { env "$@" timeout 400 python $R/bench.py $EXTRA --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu --steps 30 --warmup 5 2>$O/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$* $EXTRA', d['ms_per_step'], d['config'].get('final_loss'))"; }
{
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_norm_prev.so
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_norm_prev.so
EXTRA="--config sd15"
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_norm_prev.so
EXTRA=""
} 2>&1 | tee $O/step_ab.txt
timeout 900 python tools/determinism_probe.py sd15 64 4 > $O/determinism_sd15.txt 2>&1; grep -n "buffers compared" -A50 $O/determinism_sd15.txt | cut -c1-170
