# whole-step A/B of HIP runtime environment knobs (each line: the knob, ms per step)
R=$GRAFT_REPO_ROOT
run() { env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-concurrent --no-train-loop --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['ms_per_step'])"; }
run A=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1000
run DEBUG_HIP_KERNARG_COPY_OPT=0
run ROC_USE_FGS_KERNARG=0
run GPU_MAX_HW_QUEUES=1
run A=1
