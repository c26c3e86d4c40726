set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_p2_config4.py tests/test_gpu_p2.py -x -q -m gpu --durations=5 > $O/pytest.txt 2>&1 || { tail -60 $O/pytest.txt; exit 1; }
tail -12 $O/pytest.txt
