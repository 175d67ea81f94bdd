# The round's evidence in one gpurun call (round 6): kernel trace + stats of the default bench step, the two PMC passes, the merged step profile
# (profiles/r06_sdxl1024_ti_step_profile.json + per-kernel CSV), last-step tables, text-encoder phases, the variants and the default bench line.
# (every profiler / bench call under its own timeout: one PMC pass of this round hung for the whole 45-minute limit and ran in 6 s when repeated)
# usage (on the GPU box): bash tools/final_measure_r06.sh <commit>   (the step profile also records a hash of the kernel sources: bench.py's roofline.traffic_sources_match)
set -x
R=$GRAFT_REPO_ROOT
C=${1:-unknown}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/final_r06
mkdir -p $O
B="--no-cpu-baseline --no-concurrent --no-train-loop --no-sustained"
rm -rf /tmp/pf /tmp/pmc_f /tmp/pmc_w
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python $R/bench.py $B --steps 10 --warmup 3 > $O/bench_profiled.json 2>/dev/null
cp $(ls /tmp/pf/*/*kernel_stats.csv | head -1) $O/r06_sdxl1024_ti_rocprofv3_kernel_stats.csv
T=$(ls /tmp/pf/*/*kernel_trace.csv | head -1)
python $R/tools/last_step_auto.py $T 70 > $O/r06_sdxl1024_ti_last_step_kernels.txt 2>&1
python $R/tools/text_phase.py $T > $O/r06_sdxl1024_ti_text_phases.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $R/bench.py --no-graph $B --steps 2 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $R/bench.py --no-graph $B --steps 2 --warmup 1 > /dev/null 2>&1
python $R/tools/step_profile.py $T $(ls /tmp/pmc_f/*/*counter_collection.csv | head -1) $(ls /tmp/pmc_w/*/*counter_collection.csv | head -1) \
  $O/r06_sdxl1024_ti_step_profile.json $O/r06_sdxl1024_ti_per_kernel.csv $C "python bench.py $B" > /dev/null 2> $O/step_profile.err
# cfg2 (SD1.5 512 px batch 4, rank-16 LoRA + TI): kernel trace + last-step table + the two PMC passes -> its own step profile
rm -rf /tmp/pf15 /tmp/pmc15_f /tmp/pmc15_w
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf15 -- python $R/bench.py --config sd15 $B --steps 10 --warmup 3 > $O/bench_profiled_sd15.json 2>/dev/null
cp $(ls /tmp/pf15/*/*kernel_stats.csv | head -1) $O/r06_sd15_512_b4_rocprofv3_kernel_stats.csv
T15=$(ls /tmp/pf15/*/*kernel_trace.csv | head -1)
python $R/tools/last_step_auto.py $T15 70 > $O/r06_sd15_512_b4_last_step_kernels.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc15_f -- python $R/bench.py --config sd15 --no-graph $B --steps 2 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc15_w -- python $R/bench.py --config sd15 --no-graph $B --steps 2 --warmup 1 > /dev/null 2>&1
python $R/tools/step_profile.py $T15 $(ls /tmp/pmc15_f/*/*counter_collection.csv | head -1) $(ls /tmp/pmc15_w/*/*counter_collection.csv | head -1) \
  $O/r06_sd15_512_b4_step_profile.json $O/r06_sd15_512_b4_per_kernel.csv $C "python bench.py --config sd15 $B" > /dev/null 2> $O/step_profile_sd15.err
# cfg5 (SDXL 512 px batch 4 full fine-tune, AdamW8bit): last-step table
cd /tmp; rm -rf /tmp/pf5
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf5 -- python $R/bench.py --full-ft --no-cpu-baseline --steps 6 --warmup 3 > /dev/null 2>&1
python $R/tools/last_step_auto.py $(ls /tmp/pf5/*/*kernel_trace.csv | head -1) 45 > $O/r06_fullft_sdxl512_b4_last_step_kernels.txt 2>&1
cd $R
for v in "--no-ti" "--ti-frozen" "--config sd15" "--full-ft" "--rank 24" "--rank 64" "--jobs-per-gpu 2" "--dora" "--config sd15 --full-ft"; do
  timeout 600 python bench.py $v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],2), round(d['value'],2), round(d['roofline']['frac'],4))" >> $O/r06_bench_variants.txt
done
# the per-launch floor model of the step (roofline.shape_floor_ms of the bench line)
timeout 1200 python tools/step_floor.py --out $O/r06_step_floor.json --commit $C 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/r06_step_floor.txt
cp $O/r06_step_floor.json $R/profiles/r06_step_floor.json
timeout 900 python bench.py --profile-json $O/r06_sdxl1024_ti_step_profile.json > $O/r06_bench_line.json 2> $O/bench_default.err
# RCCL sanity on the one GPU of the box: the driver's launch line with one rank (process group "nccl" = RCCL, barrier + MAX reduction of the timing)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 $B > $O/r06_bench_line_torchrun_1rank.json 2> $O/torchrun.err
timeout 900 python bench.py --full-ft --dry-collectives --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | grep "^{" | tail -1 > $O/r06_fullft_dry_collectives.json
timeout 1500 bash $R/tools/gemm_fetch_ratio.sh $O/r06_gemm_fetch_ratio.txt > $O/r06_gemm_fetch_ratio.log 2>&1
