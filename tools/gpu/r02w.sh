set -e
for rep in 1 2 3; do
timeout -k 10 300 python tools/bench_kernels.py --level 8 --only "restrict (fine" 2>&1 | grep -v "^{\|amdgpu.ids"
done
timeout -k 10 300 python tools/bench_kernels.py --level 7 --only "restrict (fine" 2>&1 | grep -v "^{\|amdgpu.ids"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -x -q -k "restrict or transfer" 2>&1 | tail -1
