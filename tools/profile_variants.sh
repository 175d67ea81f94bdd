# last-step kernel tables of the other BASELINE workloads (rocprofv3 --kernel-trace; tools/last_step_auto.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-concurrent --no-train-loop --steps 6 --warmup 2"
rocprofv3 --kernel-trace --output-format csv -d /tmp/p15 -- python $R/bench.py --config sd15 $B > /dev/null 2>&1
python $R/tools/last_step_auto.py $(ls /tmp/p15/*/*kernel_trace.csv | head -1) 40 > $R/gpurun_out/last_step_sd15.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/pft -- python $R/bench.py --full-ft $B > /dev/null 2>&1
python $R/tools/last_step_auto.py $(ls /tmp/pft/*/*kernel_trace.csv | head -1) 40 > $R/gpurun_out/last_step_fullft.txt 2>&1
