set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02e; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || { tail -40 $O/pytest_gpu.txt; exit 1; }
tail -3 $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
HYTEG_HIP_APPLY_PFD=2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_args_pfd2.json 2>&1
python bench.py --no-cpu-baseline > $O/bench_default.json 2>&1
HYTEG_HIP_APPLY_PFD=2 python bench.py --no-cpu-baseline > $O/bench_default_pfd2.json 2>&1
python bench.py --no-cpu-baseline > $O/bench_default_b.json 2>&1
HYTEG_HIP_APPLY_PFD=2 python bench.py --no-cpu-baseline > $O/bench_default_pfd2_b.json 2>&1
HYTEG_BENCH_BACKEND=gloo HYTEG_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_gloo_rehearsal.json 2>&1 || echo "2-rank rehearsal failed"
tail -2 $O/bench_2ranks_gloo_rehearsal.json
for f in bench_driver_args bench_driver_args_pfd2 bench_default bench_default_pfd2 bench_default_b bench_default_pfd2_b; do tail -1 $O/$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', round(d['value']/1e9,1), round(d['ms_per_step']*1e3,2), round(r['launch_us'],2), round(r['launch_us_mean_over_timed_region'],2), round(r['frac'],3), r['kernel'])"; done
