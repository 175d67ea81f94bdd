#!/bin/bash
# per-kernel table of the full fine-tune step (cfg5) at HEAD
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/t; rm -rf /tmp/pf
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python $R/bench.py --full-ft --no-cpu-baseline --steps 6 --warmup 3 > $R/gpurun_out/t/bench_fullft.json 2>/dev/null
python $R/tools/last_step_auto.py $(ls /tmp/pf/*/*kernel_trace.csv | head -1) 45 > $R/gpurun_out/t/r05_fullft_sdxl512_b4_last_step_kernels.txt 2>&1
head -50 $R/gpurun_out/t/r05_fullft_sdxl512_b4_last_step_kernels.txt | cut -c1-180
