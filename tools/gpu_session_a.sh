#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/wsk_lab.sh
mkdir -p gpurun_out/sessA
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wsk" > gpurun_out/sessA/tests_wsk.log 2>&1; tail -3 gpurun_out/sessA/tests_wsk.log
timeout 1500 python -m pytest tests/test_ddp_gpu.py -x -q > gpurun_out/sessA/tests_ddp.log 2>&1; tail -3 gpurun_out/sessA/tests_ddp.log
