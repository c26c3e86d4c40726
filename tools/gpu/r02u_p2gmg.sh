set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02u; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_p2_gmg.py tests/test_gpu_host.py tests/test_gpu_p2.py -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -60 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
