#!/bin/bash
# round 6, final evidence: the driver's suite (timed), then tools/final_measure_r06.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/tests
( time timeout 1700 python -m pytest tests -q -m gpu -x --durations=15 ) > gpurun_out/tests/tests_gpu_r06.log 2>&1; tail -25 gpurun_out/tests/tests_gpu_r06.log
bash tools/final_measure_r06.sh ${COMMIT:-wip} > gpurun_out/final_r06.log 2>&1
cat gpurun_out/final_r06/r06_bench_variants.txt; tail -c 1500 gpurun_out/final_r06/r06_bench_line.json; head -3 gpurun_out/final_r06/r06_sdxl1024_ti_last_step_kernels.txt
