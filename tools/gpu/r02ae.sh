set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02ae; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_stokes.py tests/test_gpu_host.py tests/test_gpu_batch.py -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 600 python tools/bench_kernels.py 2>&1 | grep "V(3,3)\|Stokes"| grep -v "^{"
