set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02aj; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for f in bench_driver_args bench_default; do tail -1 $O/$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', round(d['value']/1e9,1), round(d['ms_per_step']*1e3,2), round(r['launch_us'],2), round(r['frac'],3), r['kernel'], r['traffic'])"; done
