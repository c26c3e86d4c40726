set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02p2p; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu > $O/pytest_dist.txt 2>&1 || { tail -60 $O/pytest_dist.txt; exit 1; }
tail -3 $O/pytest_dist.txt
export HYTEG_BENCH_BACKEND=gloo HYTEG_BENCH_SHARE_GPU=1
for n in 2 4; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 300 --warmup 20 > $O/bench_p2p_$n.json 2> $O/bench_p2p_$n.err || { tail -30 $O/bench_p2p_$n.err; exit 1; }
tail -1 $O/bench_p2p_$n.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('p2p shared-gpu ranks', d['n_gpus'], 'ms_per_step', round(d['ms_per_step']*1e3,2), 'us; value', round(d['value']/1e9,1), d['config']['halo_exchange'][:60])"
done
HYTEG_BENCH_P2P=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench_hooks_2.json 2> $O/bench_hooks_2.err || { tail -30 $O/bench_hooks_2.err; exit 1; }
tail -1 $O/bench_hooks_2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('hooks shared-gpu ranks', d['n_gpus'], 'ms_per_step', round(d['ms_per_step']*1e3,2), 'us', d['config']['halo_exchange'][:60])"
