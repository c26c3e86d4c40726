set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02z; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || { tail -60 $O/pytest_gpu.txt; exit 1; }
tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for f in bench_driver_args bench_default; do tail -1 $O/$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', round(d['value']/1e9,1), round(d['ms_per_step']*1e3,2), round(r['launch_us'],2), round(r['frac'],3), r['kernel'], r['traffic'], d['cpu_baseline']['value']/1e9, d['cpu_baseline']['one_cell_per_thread']['value']/1e9)"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || echo "stats pass failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_table -- python3 $R/tools/bench_kernels.py --reps 50 > $O/kernel_table_under_rocprof.txt 2>&1 || echo "table stats pass failed"
find $O -name "*kernel_stats.csv" | head
cd $R
timeout -k 10 600 python tools/bench_kernels.py > $O/kernel_table.txt 2>&1 || { tail -30 $O/kernel_table.txt; exit 1; }
grep -v "^{\|amdgpu.ids" $O/kernel_table.txt
