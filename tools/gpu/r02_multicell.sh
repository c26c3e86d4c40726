set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02mc; mkdir -p $O
export HYTEG_BENCH_MESH=pyramid_2el
for v in 0 2 0 2; do
HYTEG_AMD_SIDE_STREAM=$v python3 bench.py --no-cpu-baseline --steps 500 --warmup 20 2> $O/bench_2cells.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('side stream $v: 2 cells one rank: us per apply', round(d['ms_per_step']*1e3,2))"
done
