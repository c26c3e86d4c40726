#!/bin/bash
# bench line with the level-9 supporting figure
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_bench_l9; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
tail -1 $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('launch_us', r['launch_us'], 'frac', r['frac'], 'copy', r['copy_us']); print(r['level9_supporting']); print(r['infinity_cache_assisted']['launch_us'])"
