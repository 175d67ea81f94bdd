#!/bin/bash
# A/B of the attention kernels on the SDXL shapes in ONE gpurun call: 16-rows-per-wave (attn.hip) vs 32-rows-per-wave (attn32.hip) with 1 / 2 / 4 wave groups.
# usage: tools/attn32_probe.sh [outfile]
out=${1:-gpurun_out/attn32_probe.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
run() { echo "== $*" >> "$out"; env "$@" timeout 300 python tools/attn_probe.py >> "$out" 2>&1; }
run SDLT_ATTN_R32=0
run SDLT_ATTN32_KS_FWD=1 SDLT_ATTN32_KS_BWD=1
run SDLT_ATTN32_KS_FWD=2 SDLT_ATTN32_KS_BWD=2
run SDLT_ATTN32_KS_FWD=4 SDLT_ATTN32_KS_BWD=4
cat "$out"
