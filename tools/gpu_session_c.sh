#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/sessC
mkdir -p $O
hipcc --offload-arch=gfx950 -O2 tools/ubench/clockcal.hip -o /tmp/clockcal 2>/dev/null && /tmp/clockcal | tail -3 | tee $O/clockcal.txt
{
echo "== before the LoRA-up prefetch moved (trace0), packed"; SDLT_KERNEL_LIB=$R/tools/labship/lib_wsktrace0.so timeout 300 python tools/wsk_trace.py 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "== now, packed"; SDLT_KERNEL_LIB=$R/tools/labship/lib_wsktrace.so timeout 300 python tools/wsk_trace.py 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "== now, row-major"; SDLT_WSK_PACK=0 SDLT_KERNEL_LIB=$R/tools/labship/lib_wsktrace.so timeout 300 python tools/wsk_trace.py 2>&1 | grep -v "Warning\|amdgpu.ids"
} | tee $O/trace.txt
timeout 300 python tools/wsk_pack_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/probe.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wsk" > $O/tests_wsk.log 2>&1; tail -3 $O/tests_wsk.log
timeout 900 python tools/determinism_probe.py sdxl 128 1 > $O/determinism.txt 2>&1; grep -n "first differing" -A70 $O/determinism.txt | head -90
