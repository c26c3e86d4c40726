#!/bin/bash
# A/B of apply-kernel variants selected by environment switches: parity first, then alternating bench.py runs.
# usage: r03_apply_ab.sh <tag> <ENVVAR> [pairs]
set -e
TAG=${1:-ab}; VAR=${2:-HYTEG_HIP_APPLY_ALIGNED}; PAIRS=${3:-3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_$TAG; mkdir -p $O
env $VAR=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_levels.py -x -q -m gpu -k "apply" > $O/parity_variant.log 2>&1 || { tail -30 $O/parity_variant.log; exit 1; }
tail -2 $O/parity_variant.log
for i in $(seq 1 $PAIRS); do
  for v in 0 1; do
    env $VAR=$v python bench.py --no-cpu-baseline > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err || { tail -5 $O/bench_${v}_$i.err; exit 1; }
    tail -1 $O/bench_${v}_$i.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$VAR=$v run $i', 'launch_us', round(r['launch_us'],3), 'min', round(r['launch_us_min_region'],3), 'first', round(r['launch_us_first_region'],3), 'frac', round(r['frac'],4), 'copy_us', round(r['copy_us'],3), 'frac_of_copy', round(r['frac_of_copy'],3))"
  done
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_args.json 2> $O/bench_driver_args.err
tail -1 $O/bench_driver_args.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('driver args', 'launch_us', round(r['launch_us'],3), 'min', round(r['launch_us_min_region'],3), 'first', round(r['launch_us_first_region'],3), 'frac', round(r['frac'],4), 'copy_us', round(r['copy_us'],3), 'ms_per_step', d['ms_per_step'], d['regions'])"
