set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02batchsor; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "shared_between or vcycle" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
