# rocprofv3 kernel trace of the default bench step -> last-step table + text-encoder phases under gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT
TAG=${1:-prof}
shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $R/bench.py --no-cpu-baseline --no-concurrent --no-train-loop --steps 8 --warmup 3 "$@" > $R/gpurun_out/$TAG/bench.json 2> /dev/null
cp $(ls /tmp/prof_$TAG/*/*kernel_stats.csv | head -1) $R/gpurun_out/$TAG/kernel_stats.csv
T=$(ls /tmp/prof_$TAG/*/*kernel_trace.csv | head -1)
python $R/tools/last_step_auto.py $T 70 > $R/gpurun_out/$TAG/last_step.txt 2>&1
python $R/tools/text_phase.py $T > $R/gpurun_out/$TAG/text_phase.txt 2>&1
python $R/tools/step_gaps.py $T 25 > $R/gpurun_out/$TAG/gaps.txt 2>&1
