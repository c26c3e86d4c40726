set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02r; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || { tail -60 $O/pytest_gpu.txt; exit 1; }
tail -3 $O/pytest_gpu.txt
timeout -k 10 600 python tools/bench_kernels.py > $O/kernel_table.txt 2>&1 || { tail -30 $O/kernel_table.txt; exit 1; }
grep -v "^{\|amdgpu.ids" $O/kernel_table.txt
