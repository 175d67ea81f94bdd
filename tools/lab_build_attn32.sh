#!/bin/bash
# A/B kernel lab for attn32.hip: builds variants (-D switches) into tools/lab/lib_<name>.so; select with SDLT_KERNEL_LIB.
# usage: tools/lab_build_attn32.sh name1:"-DFLAG ..." name2:"..."      (the other objects are the in-tree ones: run `make` first)
cd "$(dirname "$0")/../sd-lora-trainer_amd/csrc" || exit 1
mkdir -p ../../tools/lab
TL=$(python3 -c "import os,torch;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
OTHERS="gemm.o attn.o norm.o elementwise.o lora_grad.o ti.o optim.o wgrad.o dora.o strip.o wsk.o daam.o capi.o"
build() { # name, flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form $2 -c ${SRC:-attn32.hip} -o ../../tools/lab/attn32_$1.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../tools/lab/attn32_$1.o $OTHERS -o ../../tools/lab/lib_$1.so -L$TL -Wl,-rpath,$TL && echo built $1
}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  [ "$flags" = "$v" ] && flags=""
  build "$name" "$flags" &
done
wait
