# A/B of one environment switch on the default bench step, interleaved rounds in ONE session: bash tools/env_ab2.sh VAR a b [bench args]
R=$GRAFT_REPO_ROOT
V=$1; A=$2; Bv=$3; shift; shift; shift
O=$R/gpurun_out/env_ab_$V
mkdir -p $O
cd $R
B="--no-cpu-baseline --no-concurrent --no-train-loop --steps 30 --warmup 5"
for round in 1 2 3; do
  for x in $A $Bv; do
    env $V=$x timeout 600 python bench.py $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$x', round(d['ms_per_step'],3), d['config'].get('final_loss'))" | tee -a $O/ab.txt
  done
done
