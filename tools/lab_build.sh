#!/bin/bash
# A/B kernel lab: builds variants of gemm.hip (-D switches) into tools/lab/lib_<name>.so; select with SDLT_KERNEL_LIB.
cd "$(dirname "$0")/../sd-lora-trainer_amd/csrc" || exit 1
TL=$(python3 -c "import os,torch;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
build() { # name, flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $2 -c gemm.hip -o ../../tools/lab/gemm_$1.o 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../tools/lab/gemm_$1.o attn.o norm.o elementwise.o lora_grad.o ti.o capi.o -o ../../tools/lab/lib_$1.so -L$TL -Wl,-rpath,$TL && echo built $1
}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  build "$name" "$flags" &
done
wait
