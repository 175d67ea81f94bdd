#!/bin/bash
# round 6, session A (diagnostics): the driver's bench line on the cold box first, telemetry probe, the same line warm, attn32 clock stamps, attention probe
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06a
mkdir -p $O
B="--no-cpu-baseline --no-concurrent --no-train-loop"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $B 2>$O/bench_cold.err | tail -1 > $O/bench_cold.json
timeout 300 python tools/telemetry_probe.py > $O/telemetry_probe.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 > $O/bench_warm1.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 > $O/bench_warm2.json
SDLT_KERNEL_LIB=$R/sd-lora-trainer_amd/liblab_trace.so timeout 300 python tools/attn32_trace.py > $O/attn32_trace.txt 2>&1
timeout 300 python tools/attn_probe.py > $O/attn_probe.txt 2>&1
for f in bench_cold bench_warm1 bench_warm2; do python - $O/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['ms_per_step'],3), d.get('telemetry'), (d.get('sustained') or {}).get('ms_per_step'), (d.get('sustained') or {}).get('telemetry'))
PY
done
cat $O/telemetry_probe.txt | tail -12
cat $O/attn_probe.txt | grep -v Warning | tail -8
