#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/sessD
mkdir -p $O
{
echo "== packed, separate launches"; SDLT_KERNEL_LIB=$R/tools/labship/lib_wsktrace.so timeout 300 python tools/wsk_trace.py 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "== packed, second of two back-to-back launches"; WSK_TRACE_DOUBLE=1 SDLT_KERNEL_LIB=$R/tools/labship/lib_wsktrace.so timeout 300 python tools/wsk_trace.py 2>&1 | grep -v "Warning\|amdgpu.ids"
} | tee $O/trace.txt
timeout 300 python tools/wsk_pack_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/probe.txt
SDLT_KERNEL_LIB=$R/tools/labship/lib_wsk_prev.so timeout 300 python tools/wsk_pack_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/probe.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wsk or groupnorm" > $O/tests_wsk.log 2>&1; tail -3 $O/tests_wsk.log
run() { env "$@" timeout 400 python $R/bench.py --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu --steps 30 --warmup 5 2>$O/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['config'].get('final_loss'))"; }
{
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_wsk_prev.so
run SDLT_WSK_PACK=0 SDLT_KERNEL_LIB=$R/tools/labship/lib_wsk_r04.so
run A=new
run SDLT_KERNEL_LIB=$R/tools/labship/lib_wsk_prev.so
} 2>&1 | tee $O/step_ab.txt
timeout 900 python tools/determinism_probe.py sdxl 128 1 > $O/determinism.txt 2>&1; grep -n "buffers compared" -A12 $O/determinism.txt | cut -c1-160
