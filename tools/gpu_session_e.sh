#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/sessE
mkdir -p $O
run() { env "$@" timeout 400 python $R/bench.py --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu --steps 30 --warmup 5 2>$O/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['config'].get('final_loss'))"; }
{
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_gemm_prev.so
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_gemm_prev.so
} 2>&1 | tee $O/step_ab.txt
timeout 3000 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; tail -5 $O/tests_gpu.log
