set -e
timeout -k 10 900 python -m pytest tests/test_gpu_p2_transfer.py tests/test_gpu_p2_gmg.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python tools/bench_kernels.py --level 8 --only "P2 restrict" 2>&1 | grep -v "^{\|amdgpu.ids"
timeout -k 10 300 python tools/bench_kernels.py --level 6 --only "P2 restrict" 2>&1 | grep -v "^{\|amdgpu.ids"
