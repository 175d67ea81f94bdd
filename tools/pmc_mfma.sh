# MFMA / CU busy counters per kernel family over the last eager step (two PMC passes, --kernel-trace only)
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="--no-graph --no-cpu-baseline --no-concurrent --no-train-loop --no-library-gpu --no-families --no-sustained --steps 2 --warmup 1"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pa -- python $R/bench.py $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/pb -- python $R/bench.py $B > /dev/null 2>&1
cd $R
python tools/pmc_family.py gpurun_out/pmc_mfma_family.json $(ls /tmp/pa/*/*counter_collection.csv | head -1) $(ls /tmp/pb/*/*counter_collection.csv | head -1) > /dev/null
