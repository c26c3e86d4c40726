set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02k; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py tests/test_gpu_batch.py tests/test_gpu_large_levels.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -70 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
python tools/bench_kernels.py > $O/kernel_table.txt 2>&1 || tail -20 $O/kernel_table.txt
grep -v "^{" $O/kernel_table.txt
