#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/sessI
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $O/tests_attn.log 2>&1; tail -4 $O/tests_attn.log
timeout 900 python tools/determinism_probe.py sd15 64 4 > $O/determinism_sd15.txt 2>&1; grep -n "buffers compared" -A6 $O/determinism_sd15.txt | cut -c1-170
timeout 1500 python -m pytest tests/test_real_topology_gpu.py -x -q -k "full-size and not parity" > $O/tests_full.log 2>&1; tail -4 $O/tests_full.log
timeout 400 python bench.py --config sd15 --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sd15', d['ms_per_step'], d['config'].get('final_loss'))"
timeout 400 python bench.py --config sd15 --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sd15', d['ms_per_step'], d['config'].get('final_loss'))"
