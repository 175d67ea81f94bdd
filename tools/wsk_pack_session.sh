#!/bin/bash
# Round 5, first GPU session: parity of the packed-weight wave-split-K kernels, their hot-loop timings (ring depth 2 / 3 / 4, staggered refills on / off, round-4 kernels),
# and the whole-step A/B (bench.py) of the packed path against the row-major one and against round 4's wave-split-K object.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/wskpack
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wsk" > $O/tests.log 2>&1; tail -3 $O/tests.log
for r in 3 2 4; do SDLT_WSK_WP_R=$r timeout 300 python tools/wsk_pack_probe.py 2>&1 | grep -v Warning; done | tee $O/probe.txt
SDLT_WSK_WP_R=3 SDLT_WSK_STAGGER=0 timeout 300 python tools/wsk_pack_probe.py 2>&1 | grep -v Warning | tee -a $O/probe.txt
SDLT_WSK_PACK=0 SDLT_KERNEL_LIB=$R/tools/labship/lib_wsk_r04.so timeout 300 python tools/wsk_pack_probe.py 2>&1 | grep -v Warning | tee -a $O/probe.txt
run() { env "$@" timeout 400 python $R/bench.py --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu --steps 30 --warmup 5 2>$O/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'], d['config'].get('final_loss'))"; }
{
run SDLT_WSK_PACK=1
run SDLT_WSK_PACK=0
run SDLT_WSK_PACK=0 SDLT_KERNEL_LIB=$R/tools/labship/lib_wsk_r04.so
run SDLT_WSK_PACK=1 SDLT_WSK_WP_R=2
run SDLT_WSK_PACK=1 SDLT_WSK_WP_R=4
run SDLT_WSK_PACK=1
run SDLT_WSK_PACK=0 SDLT_KERNEL_LIB=$R/tools/labship/lib_wsk_r04.so
} 2>&1 | tee $O/step_ab.txt
