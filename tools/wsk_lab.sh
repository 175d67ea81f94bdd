#!/bin/bash
# What bounds the K walk of the wave-split-K GEMM?  Lab build (-DSDLT_WSK_LAB) of wsk.hip: SDLT_WSK_STAGGER bit 1 = every workgroup reads row tile 0 of X, bit 2 = column tile 0 of W
# (that operand is then L2-resident in every XCD: no fabric traffic, same L2 -> CU bytes), bit 3 = no MFMAs (packed kernels).  Results are garbage; timings per launch, graph-replayed, rotating weights.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/wsklab
mkdir -p $O
export SDLT_KERNEL_LIB=$R/tools/labship/lib_wsklab.so
for st in 1 3 5 7 9 15; do SDLT_WSK_STAGGER=$st timeout 300 python tools/wsk_pack_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee $O/lab.txt
