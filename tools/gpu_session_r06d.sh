#!/bin/bash
# round 6, session D: attention tests + 16-byte store A/B, DDP tests (one process pair), trained-like parity case, the per-launch floor model
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06d
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or attn or wsk or rank_groups or packed" > $O/tests_attn.log 2>&1; tail -3 $O/tests_attn.log
timeout 900 python -m pytest tests/test_ti_step_gpu.py tests/test_step_gpu.py -q -x -k "dora or text_encoder_lora or baseline_config" > $O/tests_dora.log 2>&1; tail -3 $O/tests_dora.log
timeout 900 python -m pytest tests/test_real_topology_gpu.py -q -x -k "dora-step or rank24 or rank64" > $O/tests_real_dora.log 2>&1; tail -3 $O/tests_real_dora.log
( time timeout 900 python -m pytest tests/test_ddp_gpu.py -q -x ) > $O/tests_ddp.log 2>&1; tail -6 $O/tests_ddp.log
timeout 1200 python -m pytest tests/test_real_topology_gpu.py -q -x -k "trained-like" > $O/tests_trained_like.log 2>&1; tail -30 $O/tests_trained_like.log
cp gpurun_out/parity_report.json $O/parity_report_trained_like.json 2>/dev/null
python tools/attn_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/attn_probe_new.txt
SDLT_KERNEL_LIB=$R/sd-lora-trainer_amd/liblab_st8.so python tools/attn_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/attn_probe_st8.txt
paste -d'\n' $O/attn_probe_new.txt $O/attn_probe_st8.txt
bash tools/lib_ab.sh st8 > $O/lib_ab_st8.log 2>&1; cat gpurun_out/lib_ab_st8/ab.txt
B="--no-cpu-baseline --no-concurrent --no-train-loop --no-sustained --steps 30 --warmup 5"
ab() { L=$1; EV=$2; shift; shift
  env $EV timeout 600 python bench.py $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', '$EV', round(d['ms_per_step'],3), 'loss', d['config']['final_loss'])" | tee -a $O/ab.txt
}
for rep in 1 2; do
  ab "rank24" "SDLT_WSK_RANKS=16" --rank 24
  ab "rank24" "SDLT_WSK_RANKS=16,32" --rank 24
  ab "dora" "SDLT_WSK_DORA=0" --dora
  ab "dora" "SDLT_WSK_DORA=1" --dora
  ab "rank64" "SDLT_X=1" --rank 64
done
timeout 1200 python tools/step_floor.py --out $O/r06_step_floor.json --commit ${COMMIT:-wip} > $O/step_floor.txt 2>&1; grep -v "Warning\|amdgpu.ids" $O/step_floor.txt | tail -40
