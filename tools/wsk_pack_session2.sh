#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/wskpack
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wsk" > $O/tests.log 2>&1; tail -3 $O/tests.log
for r in 3 2 4; do SDLT_WSK_WP_R=$r timeout 300 python tools/wsk_pack_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee $O/probe.txt
SDLT_WSK_WP_R=3 SDLT_WSK_STAGGER=0 timeout 300 python tools/wsk_pack_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/probe.txt
