#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/sessF
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wsk or conv" > $O/tests_wsk.log 2>&1; tail -4 $O/tests_wsk.log
timeout 600 python tools/wsk_conv_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/conv_probe.txt
run() { env "$@" timeout 400 python $R/bench.py --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu --steps 30 --warmup 5 2>$O/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['config'].get('final_loss'))"; }
{
run SDLT_WSK_CONV=1
run SDLT_WSK_CONV=0
run SDLT_WSK_CONV=1
run SDLT_WSK_CONV=0
} 2>&1 | tee $O/step_ab.txt
timeout 900 python bench.py --full-ft --dry-collectives --no-cpu-baseline --steps 5 --warmup 2 > $O/dry_collectives.json 2>$O/dry_err.log; tail -c 1500 $O/dry_collectives.json; tail -3 $O/dry_err.log
timeout 1500 python -m pytest tests/test_real_topology_gpu.py -x -q -k "trajectory" > $O/tests_traj.log 2>&1; tail -4 $O/tests_traj.log
