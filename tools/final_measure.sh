set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/final
B="--no-cpu-baseline --no-concurrent --no-train-loop"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python $R/bench.py $B --steps 10 --warmup 3 > /dev/null 2>&1
cp $(ls /tmp/prof_final/*/*kernel_stats.csv | head -1) $R/gpurun_out/final/kernel_stats.csv
python $R/tools/last_step_auto.py $(ls /tmp/prof_final/*/*kernel_trace.csv | head -1) 60 > $R/gpurun_out/final/last_step.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $R/bench.py --no-graph $B --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $R/bench.py --no-graph $B --steps 2 --warmup 1 > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(ls /tmp/pmc_f/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_w/*/*counter_collection.csv | head -1) $R/gpurun_out/final/traffic.json > /dev/null 2>&1
cp $R/gpurun_out/final/traffic.json $R/profiles/r02_sdxl1024_ti_hbm_traffic_pmc.json
cd $R
for v in "--no-ti" "--ti-frozen" "--config sd15" "--full-ft" "--rank 64" "--jobs-per-gpu 2" "--dora" "--config sd15 --full-ft"; do
  python bench.py $v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d['roofline']['frac'])" >> gpurun_out/final/variants.txt
done
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
