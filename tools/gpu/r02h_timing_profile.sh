set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fp32.py tests/test_gpu_timing.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -60 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
bash tools/profile_bench.sh > $O/profile_bench.log 2>&1 || { tail -30 $O/profile_bench.log; }
tail -5 $O/profile_bench.log
