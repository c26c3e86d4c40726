set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02chain; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_host.py tests/test_gpu_faces.py tests/test_gpu_sor_shell.py tests/test_gpu_distributed.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -60 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
bash tools/gpu/r02_rank_probe.sh
