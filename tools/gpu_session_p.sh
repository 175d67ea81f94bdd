#!/bin/bash
# attention backward's row term D as a side output of the to_out.0 dX product: kernel tests, step parity subset, A/B
cd /root/repo; mkdir -p gpurun_out/p
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "rowdot or attention_fwd_bwd or wsk" > gpurun_out/p/tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/p/tests.log
timeout 1500 python -m pytest tests/test_step_gpu.py tests/test_ti_step_gpu.py -x -q > gpurun_out/p/tests2.log 2>&1; echo "tests2 rc=$?"; tail -5 gpurun_out/p/tests2.log
for rep in 1 2 3; do
for f in 1 0; do
  SDLT_ROWDOT=$f python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('rowdot=$f',round(d['ms_per_step'],3), d.get('final_loss'))"
done
done
