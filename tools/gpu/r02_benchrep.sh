set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02zz; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || echo "stats pass failed"
cd $R
for f in bench_driver_args bench_default bench_under_rocprof; do tail -1 $O/$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', round(d['value']/1e9,1), round(d['ms_per_step']*1e3,2), round(r['launch_us'],2), round(r['frac'],3))"; done
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof_bench/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:2]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
