set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_stokes.py -x -q -m gpu > $O/pytest_stokes.txt 2>&1 || { tail -60 $O/pytest_stokes.txt; exit 1; }
tail -5 $O/pytest_stokes.txt
