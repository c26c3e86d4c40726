set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02p; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_host.py -m gpu -x -q -k "restrict or prolong or transfer or gmg or multigrid or batch" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
for lvl in 8 7 6 5; do
echo "== level $lvl"
timeout -k 10 300 python tools/bench_kernels.py --level $lvl --only "restrict (fine" 2>&1 | grep -v "^{\|amdgpu.ids"
timeout -k 10 300 python tools/bench_kernels.py --level $lvl --only "olongate Repl" 2>&1 | grep -v "^{\|amdgpu.ids"
HYTEG_HIP_PROL_NOSHELL=1 timeout -k 10 300 python tools/bench_kernels.py --level $lvl --only "olongate Repl" 2>&1 | grep -v "^{\|amdgpu.ids"
done
