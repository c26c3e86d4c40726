set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02y; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fp32.py -m gpu -x -q -k "apply or jacobi or Jacobi or fp32 or f32" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for rep in 1 2; do
for lvl in 8; do
echo "== level $lvl"
timeout -k 10 300 python tools/bench_kernels.py --level $lvl --only "apply" 2>&1 | grep -v "^{\|amdgpu.ids\|P2\|P1P1"
timeout -k 10 300 python tools/bench_kernels.py --level $lvl --only "Jacobi" 2>&1 | grep -v "^{\|amdgpu.ids"
done
done
