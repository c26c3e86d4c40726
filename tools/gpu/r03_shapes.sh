#!/bin/bash
# bench.py for a list of experimental brick shapes (HYTEG_HIP_APPLY_SHAPE); parity of every shape first (level 8 and 9)
# usage: r03_shapes.sh <tag> shape1 shape2 ...   ("default" = no switch)
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_$TAG; mkdir -p $O
for sh in "$@"; do
  if [ $sh != default ]; then export HYTEG_HIP_APPLY_SHAPE=$sh; else unset HYTEG_HIP_APPLY_SHAPE; fi
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level8_full_size or random_weights" > $O/parity_$sh.log 2>&1 || { echo "PARITY FAILED $sh"; tail -15 $O/parity_$sh.log; continue; }
  for i in 1 2; do
    python bench.py --no-cpu-baseline --regions 11 > $O/bench_${sh}_$i.json 2> $O/bench_${sh}_$i.err || { echo "bench failed $sh"; tail -5 $O/bench_${sh}_$i.err; continue; }
    tail -1 $O/bench_${sh}_$i.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$sh run $i', 'launch_us', round(r['launch_us'],3), 'min', round(r['launch_us_min_region'],3), 'frac', round(r['frac'],4), 'copy_us', round(r['copy_us'],3), 'frac_of_copy', round(r['frac_of_copy'],3))"
  done
done
