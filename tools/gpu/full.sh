#!/bin/bash
# whole GPU suite + the two bench lines + smoke.  usage: full.sh [tag]
TAG=${1:-full}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_$TAG; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err || tail -5 $O/bench_driver_args.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
for f in bench_driver_args bench_default; do tail -1 $O/$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', 'GDoF/s', round(d['value']/1e9,1), 'us/step', round(d['ms_per_step']*1e3,2), 'launch_us', round(r['launch_us'],3), 'frac', round(r['frac'],4), 'copy_us', round(r['copy_us'],3), 'frac_of_copy', round(r['frac_of_copy'],3), 'cpu', d.get('cpu_baseline',{}).get('value'))"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
# rehearsal of the driver's multi-GPU command line on this box's ONE GPU: two ranks share it, gloo carries the hooks
HYTEG_BENCH_BACKEND=gloo HYTEG_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err \
  && tail -1 $O/bench_two_ranks_one_gpu.json | cut -c1-300 || tail -5 $O/bench_two_ranks_one_gpu.err
