# A/B of two builds of the kernel library in ONE session on one box: the in-tree libsdlt_kernels.so against sd-lora-trainer_amd/liblab_<name>.so
# (tools/lab_build_obj.sh <file.hip> name:"-DFLAG" builds tools/lab/lib_<name>.so; copy it next to the package's library so that it travels).
# usage (GPU box): bash tools/lib_ab.sh <name> [bench args of a second workload]
R=$GRAFT_REPO_ROOT
N=$1; shift
O=$R/gpurun_out/lib_ab_$N
mkdir -p $O
cd $R
B="--no-cpu-baseline --no-concurrent --no-train-loop --steps 30 --warmup 5"
run() { # label, lib ("" = in-tree), extra bench args
  L=$1; LIB=$2; shift; shift
  SDLT_KERNEL_LIB=$LIB timeout 600 python bench.py $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['ms_per_step'],3))" | tee -a $O/ab.txt
}
LAB=$R/sd-lora-trainer_amd/liblab_$N.so
for round in 1 2 3; do
  run "in-tree" ""
  run "$N" $LAB
done
for w in "--config sd15" "--full-ft" "$@"; do
  [ -z "$w" ] && continue
  run "in-tree $w" "" $w
  run "$N $w" $LAB $w
done
