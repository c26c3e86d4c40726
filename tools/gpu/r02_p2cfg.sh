set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02p2p; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_stokes_distributed.py -x -q -m gpu > $O/pytest_stokes_dist.txt 2>&1 || { tail -50 $O/pytest_stokes_dist.txt; exit 1; }
tail -2 $O/pytest_stokes_dist.txt
