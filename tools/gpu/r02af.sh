set -e
timeout -k 10 900 python -m pytest tests/test_gpu_p2.py tests/test_gpu_p2_gmg.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python tools/bench_p2_apply.py --levels 6 7 2>&1 | grep -v "amdgpu.ids"
