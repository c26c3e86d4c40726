set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02vc; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fp32.py -x -q -m gpu > $O/pytest3.txt 2>&1 || { tail -40 $O/pytest3.txt; exit 1; }
tail -2 $O/pytest3.txt
timeout -k 10 600 python tools/bench_kernels.py --reps 10 2>&1 | grep "V(3,3) Jacobi"
