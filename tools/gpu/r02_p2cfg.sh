set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02p2p; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_p2_config4.py -x -q -m gpu > $O/pytest_cfg4.txt 2>&1 || { tail -40 $O/pytest_cfg4.txt; exit 1; }
tail -2 $O/pytest_cfg4.txt
