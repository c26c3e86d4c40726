set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02batchsor; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_host.py tests/test_gpu_sor_shell.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 300 python tools/vcycle_breakdown.py 2>&1 | grep -v amdgpu.ids | tail -7
timeout -k 10 600 python tools/bench_kernels.py --reps 20 2>&1 | grep "V(3,3)"
