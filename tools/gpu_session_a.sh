#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/a
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -k "groupnorm or adamw or step or optimizer" > gpurun_out/a/tests.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/a/tests.log | cut -c1-200
for rep in 1 2 3; do
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new',round(d['ms_per_step'],3))"
  SDLT_KERNEL_LIB=/root/repo/tools/labship/lib_prev.so python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev',round(d['ms_per_step'],3))"
done
