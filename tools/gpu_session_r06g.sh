#!/bin/bash
# round 6, session G: next-weight prefetch in the step - tests that capture graphs, bit-identity of losses, A/B in one session
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06g
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "wsk or rank_groups or packed" > $O/tests_wsk.log 2>&1; tail -3 $O/tests_wsk.log
timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_ti_step_gpu.py -q -x > $O/tests_step.log 2>&1; tail -3 $O/tests_step.log
timeout 900 python -m pytest tests/test_real_topology_gpu.py -q -x -k "sdxl-full-size or sdxl-step-trajectory" > $O/tests_real.log 2>&1; tail -3 $O/tests_real.log
B="--no-cpu-baseline --no-concurrent --no-train-loop --no-sustained --steps 30 --warmup 5"
ab() { L=$1; EV=$2; shift; shift
  env $EV timeout 600 python bench.py $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', '$EV', round(d['ms_per_step'],3), 'loss', d['config']['final_loss'])" | tee -a $O/ab.txt
}
for rep in 1 2 3; do
  ab "default" "SDLT_WSK_PREFETCH=0"
  ab "default" "SDLT_WSK_PREFETCH=1"
done
ab "default S=20" "SDLT_WSK_PREFETCH_STEPS=20"
ab "default S=80" "SDLT_WSK_PREFETCH_STEPS=80"
ab "no-ti" "SDLT_WSK_PREFETCH=0" --no-ti
ab "no-ti" "SDLT_WSK_PREFETCH=1" --no-ti
ab "rank24" "SDLT_WSK_PREFETCH=0" --rank 24
ab "rank24" "SDLT_WSK_PREFETCH=1" --rank 24
ab "sd15" "SDLT_WSK_PREFETCH=0" --config sd15
ab "sd15" "SDLT_WSK_PREFETCH=1" --config sd15
