set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02vc; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py tests/test_gpu_batch.py tests/test_gpu_large_levels.py tests/test_gpu_stokes.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 600 python tools/bench_kernels.py --reps 20 2>&1 | grep "V(3,3)\|^apply\|^Jacobi"
