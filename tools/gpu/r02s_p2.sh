set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02s; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_p2.py tests/test_gpu_p2_config4.py tests/test_gpu_p2_transfer.py -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
echo "== rows kernel"
timeout -k 10 300 python tools/bench_p2_apply.py 2>&1 | grep -v "amdgpu.ids"
echo "== thread-per-DoF kernel, two launches (HYTEG_HIP_P2_INNER_THREADS=1)"
HYTEG_HIP_P2_INNER_THREADS=1 timeout -k 10 300 python tools/bench_p2_apply.py 2>&1 | grep -v "amdgpu.ids"
