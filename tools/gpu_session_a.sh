#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/a
timeout 900 python -m pytest tests/test_fullft_gpu.py tests/test_kernels_gpu.py -x -q -k "fullft or affine" > gpurun_out/a/tests.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/a/tests.log | cut -c1-200
for rep in 1 2 3; do
  python bench.py --full-ft --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new',round(d['ms_per_step'],3))"
  SDLT_KERNEL_LIB=/root/repo/tools/labship/lib_prev.so python bench.py --full-ft --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev',round(d['ms_per_step'],3))"
done
