set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02dot; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_host.py tests/test_gpu_large_levels.py -x -q -m gpu -k "dot or cg or CG or gmg or solver or batch" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for l in 3 4 5 7 8; do timeout -k 10 300 python tools/bench_kernels.py --level $l --only "dot" 2>&1 | grep "^dot" ; done
