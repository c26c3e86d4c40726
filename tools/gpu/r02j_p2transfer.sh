set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02j; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_p2_transfer.py tests/test_gpu_p2_config4.py tests/test_gpu_parity.py tests/test_gpu_host.py tests/test_gpu_batch.py tests/test_gpu_large_levels.py -x -q -m gpu --durations=5 > $O/pytest.txt 2>&1 || { tail -70 $O/pytest.txt; exit 1; }
tail -12 $O/pytest.txt
python tools/bench_kernels.py > $O/kernel_table.txt 2>&1 || tail -20 $O/kernel_table.txt
cat $O/kernel_table.txt
