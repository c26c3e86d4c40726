#!/bin/bash
# bench.py under a list of environment settings, parity of the apply kernel first for each.
# usage: r03_env_ab.sh <tag> "VAR=VAL[,VAR2=VAL2]" ... ("default" = nothing set)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_$TAG; mkdir -p $O
for setting in "$@"; do
  ENVS=""
  if [ "$setting" != default ]; then ENVS=$(echo $setting | tr "," " "); fi
  name=$(echo $setting | tr "=," "__")
  env $ENVS timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level8_full_size or random_weights or brick_shape" > $O/parity_$name.log 2>&1 || { echo "PARITY FAILED $setting"; tail -15 $O/parity_$name.log; continue; }
  for i in 1 2; do
    env $ENVS python bench.py --no-cpu-baseline --regions 11 > $O/bench_${name}_$i.json 2> $O/bench_${name}_$i.err || { echo "bench failed $setting"; tail -5 $O/bench_${name}_$i.err; continue; }
    tail -1 $O/bench_${name}_$i.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$setting run $i', 'launch_us', round(r['launch_us'],3), 'min', round(r['launch_us_min_region'],3), 'frac', round(r['frac'],4), 'copy_us', round(r['copy_us'],3))"
  done
done
