#!/bin/bash
# round 6, session E: conv tile probe (128 x 80 tiles without a K split), the per-launch floor model, the trained-like parity cases (also at the full size), the 1024 px image report
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06e
mkdir -p $O
timeout 900 python tools/conv_tile_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/conv_tile_probe.txt; cat $O/conv_tile_probe.txt
timeout 1200 python tools/step_floor.py --out $O/r06_step_floor.json --commit ${COMMIT:-wip} > $O/step_floor.txt 2>&1; grep -v "Warning\|amdgpu.ids" $O/step_floor.txt | tail -45
SDLT_PARITY_EXTRA=1 timeout 1500 python -m pytest tests/test_real_topology_gpu.py -q -x -k "trained-like" > $O/tests_trained_like.log 2>&1; tail -5 $O/tests_trained_like.log
cp gpurun_out/parity_report.json $O/parity_report_trained_like.json 2>/dev/null
( time SDLT_E2E_1024=1 timeout 2400 python -m pytest tests/test_e2e_image_gpu.py -q -x -k "1024px" ) > $O/tests_e2e_1024.log 2>&1; tail -8 $O/tests_e2e_1024.log
cp gpurun_out/parity_report_e2e.json $O/parity_report_e2e_1024.json 2>/dev/null
