#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/tests
timeout 3300 python -m pytest tests -q -m gpu > gpurun_out/tests/tests_gpu.log 2>&1; tail -6 gpurun_out/tests/tests_gpu.log
bash tools/final_measure_r05.sh e0df7b9 > gpurun_out/final_r05.log 2>&1
cat gpurun_out/final_r05/r05_bench_variants.txt; tail -c 600 gpurun_out/final_r05/r05_bench_line.json
