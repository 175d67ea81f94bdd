# A/B of the folded LayerNorm in ONE session on one box.  SDLT_LN_FOLD bit mask: 1 norm1 -> q|k|v, 2 norm2 -> attn2.to_q, 4 norm3 -> ff.net.0.proj;
# SDLT_LN_PARTS=0: no row partials from the producing GEMMs (every folded consumer computes its statistics in the K walk).
# The kernel tests of the fold first, then the default bench step per setting (two rounds, interleaved), then SD1.5.
# usage (GPU box): bash tools/ln_fold_ab.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ln_fold
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "folded_layernorm or layernorm_bwd_y or row_partials" > $O/kernel_tests.log 2>&1
tail -3 $O/kernel_tests.log
B="--no-cpu-baseline --no-concurrent --no-train-loop --steps 30 --warmup 5"
run() { # label, env..., -- bench args
  L=$1; shift
  env "$@" timeout 600 python bench.py $B $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['ms_per_step'],3), d['config'].get('final_loss'))" | tee -a $O/ab.txt
}
for round in 1 2; do
  run "fold=0" SDLT_LN_FOLD=0
  run "fold=7" SDLT_LN_FOLD=7
  run "fold=7,parts=0" SDLT_LN_FOLD=7 SDLT_LN_PARTS=0
  run "fold=3" SDLT_LN_FOLD=3
  run "fold=4" SDLT_LN_FOLD=4
done
run "fold=7,width=64" SDLT_LN_FOLD=7 SDLT_LN_FOLD_WIDTH=64
EXTRA="--config sd15"
run "sd15 fold=0" SDLT_LN_FOLD=0
run "sd15 fold=7" SDLT_LN_FOLD=7
