#!/bin/bash
# AdamW8bit: kernel parity, full fine-tune tests, A/B of the optimizer pass on cfg5
cd /root/repo; mkdir -p gpurun_out/n
timeout 900 python -m pytest tests/test_adam8_gpu.py tests/test_fullft_gpu.py -x -q > gpurun_out/n/tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/n/tests.log
timeout 600 python -m pytest tests/test_step_gpu.py -x -q -k "train_generator" >> gpurun_out/n/tests2.log 2>&1; echo "tests2 rc=$?"; tail -5 gpurun_out/n/tests2.log
for rep in 1 2; do
for v in "--full-ft" "--full-ft --fp32-moments"; do
  python bench.py --steps 20 --warmup 5 $v 2>gpurun_out/n/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v',round(d['ms_per_step'],3), d.get('final_loss'), d['config']['workload'][:140])"
done
done
tail -5 gpurun_out/n/bench.err
