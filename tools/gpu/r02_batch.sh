set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02batch; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_host.py tests/test_gpu_stokes.py tests/test_gpu_sor_shell.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -60 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 900 python tools/bench_kernels.py --reps 30 > $O/table.txt 2>&1 || { tail -30 $O/table.txt; exit 1; }
grep "V(3,3)\|Stokes\|cube" $O/table.txt
