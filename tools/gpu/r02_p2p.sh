set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02p2p; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_p2_config4.py -x -q -m gpu > $O/pytest_dist.txt 2>&1 || { tail -60 $O/pytest_dist.txt; exit 1; }
tail -3 $O/pytest_dist.txt
export HYTEG_BENCH_BACKEND=gloo HYTEG_BENCH_SHARE_GPU=1
run() { # name, nranks, extra env...
  name=$1; n=$2; shift; shift
  env "$@" timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus $n --steps 300 --warmup 20 > $O/bench_$name.json 2> $O/bench_$name.err || { tail -30 $O/bench_$name.err; exit 1; }
  tail -1 $O/bench_$name.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name', 'us per step', round(d['ms_per_step']*1e3,2), d['config']['halo_exchange'][:40])"
}
run default_2 2 A=1
run default_4 4 A=1
