#!/bin/bash
# A/B kernel lab for ONE source of csrc/: tools/lab_build_obj.sh <file.hip> name1:"-DFLAG ..." ...  -> tools/lab/lib_<name>.so (select with SDLT_KERNEL_LIB);
# the other objects are the in-tree ones (run `make` first)
cd "$(dirname "$0")/../sd-lora-trainer_amd/csrc" || exit 1
mkdir -p ../../tools/lab
SRC=$1; shift
BASE=${SRC%.hip}
TL=$(python3 -c "import os,torch;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
OTHERS=$(for o in gemm attn attn32 norm elementwise lora_grad ti optim wgrad dora strip wsk daam capi; do [ "$o" != "$BASE" ] && echo -n "$o.o "; done)
EXTRA=""; case "$BASE" in attn|attn32) EXTRA="-mllvm -amdgpu-mfma-vgpr-form";; esac
build() { # name, flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $EXTRA $2 -c $SRC -o ../../tools/lab/${BASE}_$1.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../tools/lab/${BASE}_$1.o $OTHERS -o ../../tools/lab/lib_$1.so -L$TL -Wl,-rpath,$TL && echo built $1
}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  [ "$flags" = "$v" ] && flags=""
  build "$name" "$flags" &
done
wait
