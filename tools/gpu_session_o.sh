#!/bin/bash
# time-embedding column sums fused into norm2's backward: kernel test, step parity subset, A/B
cd /root/repo; mkdir -p gpurun_out/o
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "groupnorm" > gpurun_out/o/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/o/tests.log
timeout 1500 python -m pytest tests/test_step_gpu.py tests/test_ti_step_gpu.py -x -q > gpurun_out/o/tests2.log 2>&1; echo "tests2 rc=$?"; tail -5 gpurun_out/o/tests2.log
for rep in 1 2 3; do
for f in 1 0; do
  SDLT_COLSUM_FUSED=$f python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fused=$f',round(d['ms_per_step'],3), d.get('final_loss'))"
done
done
