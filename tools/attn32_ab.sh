#!/bin/bash
# A/B of attention kernel builds in ONE gpurun call: tools/attn32_ab.sh name1 name2 ...  (tools/lab/lib_<name>.so; "tree" = the in-tree library)
out=gpurun_out/attn32_ab.txt
mkdir -p gpurun_out; : > $out
for n in "$@"; do
  echo "== $n" >> $out
  if [ "$n" = tree ]; then timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids >> $out
  else SDLT_KERNEL_LIB=tools/lab/lib_$n.so timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids >> $out; fi
done
cat $out
