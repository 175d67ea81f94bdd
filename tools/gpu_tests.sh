#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/tests
timeout 3300 python -m pytest tests -q -m gpu > gpurun_out/tests/tests_gpu.log 2>&1; tail -8 gpurun_out/tests/tests_gpu.log
