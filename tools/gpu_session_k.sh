#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/sessK
mkdir -p $O
{ echo "== in-tree"; timeout 600 python tools/strip_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids"; echo "== touch"; SDLT_KERNEL_LIB=$R/tools/labship/lib_striptouch.so timeout 600 python tools/strip_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids"; } | tee $O/strip_probe.txt
run() { env "$@" timeout 400 python $R/bench.py --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu --steps 30 --warmup 5 2>$O/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['config'].get('final_loss'))"; }
{
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_striptouch.so
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_striptouch.so
} 2>&1 | tee $O/step_ab.txt
