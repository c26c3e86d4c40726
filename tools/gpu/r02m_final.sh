set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02m; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || { tail -60 $O/pytest_gpu.txt; exit 1; }
tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python hyteg_amd/csrc/exp/rccl_native_probe.py > $O/rccl_native_probe.txt 2>&1 || tail -5 $O/rccl_native_probe.txt
cat $O/rccl_native_probe.txt | grep -v amdgpu.ids
for f in bench_driver_args bench_default; do tail -1 $O/$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', round(d['value']/1e9,1), round(d['ms_per_step']*1e3,2), round(r['launch_us'],2), round(r['frac'],3), r['kernel'], r['traffic'], d['cpu_baseline']['value']/1e9, d['cpu_baseline']['one_cell_per_thread']['value']/1e9)"; done
