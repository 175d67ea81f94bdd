#!/bin/bash
# round 6, session C: the new wave-split-K paths (rank pads 32 / 64, DoRA column factor) - tests, A/B against the round-5 routing, where a rank-64 step spends its time; attn32 wave-group sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06c
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "rank_groups_and_dora or packed_weight or wsk_gemm" > $O/tests_wsk.log 2>&1; tail -3 $O/tests_wsk.log
timeout 900 python -m pytest tests/test_ti_step_gpu.py -q -x -k "text_encoder_lora" > $O/tests_te.log 2>&1; tail -3 $O/tests_te.log
B="--no-cpu-baseline --no-concurrent --no-train-loop --no-sustained --steps 30 --warmup 5"
ab() { # label "ENV=.." args...
  L=$1; EV=$2; shift; shift
  env $EV timeout 600 python bench.py $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', '$EV', round(d['ms_per_step'],3), 'loss', d['config']['final_loss'])" | tee -a $O/ab.txt
}
for rep in 1 2; do
  ab "rank64" "SDLT_WSK_RANKS=16" --rank 64
  ab "rank64" "SDLT_WSK_RANKS=16,32,64" --rank 64
  ab "rank24" "SDLT_WSK_RANKS=16" --rank 24
  ab "rank24" "SDLT_WSK_RANKS=16,32,64" --rank 24
  ab "dora" "SDLT_WSK_DORA=0" --dora
  ab "dora" "SDLT_WSK_DORA=1" --dora
done
ab "default" "SDLT_X=0"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf64
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf64 -- python $R/bench.py --rank 64 --no-cpu-baseline --no-concurrent --no-train-loop --no-sustained --steps 6 --warmup 3 > /dev/null 2>&1
python $R/tools/last_step_auto.py $(ls /tmp/pf64/*/*kernel_trace.csv | head -1) 60 > $R/$O/rank64_last_step_kernels.txt 2>&1
rm -rf /tmp/pfd
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pfd -- python $R/bench.py --dora --no-cpu-baseline --no-concurrent --no-train-loop --no-sustained --steps 6 --warmup 3 > /dev/null 2>&1
python $R/tools/last_step_auto.py $(ls /tmp/pfd/*/*kernel_trace.csv | head -1) 60 > $R/$O/dora_last_step_kernels.txt 2>&1
cd $R
bash tools/attn32_probe.sh $O/attn32_probe.txt > /dev/null 2>&1
grep -v "Warning\|amdgpu.ids" $O/attn32_probe.txt | tail -40
head -4 $O/rank64_last_step_kernels.txt
