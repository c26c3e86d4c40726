set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_host.py tests/test_gpu_large_levels.py -m gpu -x -q -k "sor or gs or gauss or SOR" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
timeout -k 10 300 python tools/bench_sor.py > $O/bench_sor.txt 2>&1 || { tail -20 $O/bench_sor.txt; exit 1; }
cat $O/bench_sor.txt
