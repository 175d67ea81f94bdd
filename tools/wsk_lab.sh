#!/bin/bash
# What bounds the K walk of the wave-split-K GEMM?  Lab build (-DSDLT_WSK_LAB) of wsk.hip: SDLT_WSK_STAGGER bit 1 = every workgroup reads row tile 0 of X, bit 2 = column tile 0 of W
# (that operand is then L2-resident in every XCD: no fabric traffic, same L2 -> CU bytes), bit 3 = no MFMAs (packed kernels).  Results are garbage; timings per launch, graph-replayed, rotating weights.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/wsklab
mkdir -p $O
mkdir -p tools/labship && bash tools/lab_build_obj.sh wsk.hip wsklab:"-DSDLT_WSK_LAB" && mv tools/lab/lib_wsklab.so tools/labship/       # (hipcc is on the GPU box too; run `make -C sd-lora-trainer_amd/csrc` first)
export SDLT_KERNEL_LIB=$R/tools/labship/lib_wsklab.so
for st in 1 3 5 7 9 15; do SDLT_WSK_STAGGER=$st timeout 300 python tools/wsk_pack_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee $O/lab.txt
