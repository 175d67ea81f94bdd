#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/tests
echo "(full GPU suite: see the previous session)"
bash tools/final_measure_r05.sh 555274e > gpurun_out/final_r05.log 2>&1
cat gpurun_out/final_r05/r05_bench_variants.txt; tail -c 600 gpurun_out/final_r05/r05_bench_line.json
