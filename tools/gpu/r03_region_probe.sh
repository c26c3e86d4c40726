#!/bin/bash
# how do short timed regions (the driver's --steps 20 --warmup 5) behave: with and without copy regions between them
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_region_probe; mkdir -p $O
for c in 1 0; do
  HYTEG_BENCH_COPY=$c python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/k20_copy$c.json 2> $O/k20_copy$c.err
  tail -1 $O/k20_copy$c.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('K=20 copy=$c', d['regions']['launch_us_all'], d['regions']['copy_us_all'])"
done
HYTEG_BENCH_COPY=1 python bench.py --steps 200 --warmup 5 --no-cpu-baseline > $O/k200.json 2> $O/k200.err
tail -1 $O/k200.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('K=200', d['regions']['launch_us_all'], d['regions']['copy_us_all'])"
