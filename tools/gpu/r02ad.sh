set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02ad; mkdir -p $O
for n in 2 4; do
HYTEG_BENCH_BACKEND=gloo HYTEG_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 20 --warmup 5 > $O/bench_n$n.json 2> $O/bench_n$n.err || { tail -20 $O/bench_n$n.err; exit 1; }
tail -1 $O/bench_n$n.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], 'value', round(d['value']/1e9,1), 'ms_per_step', round(d['ms_per_step']*1e3,1), 'us;', d['config']['halo_exchange'])"
done
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q 2>&1 | tail -2
