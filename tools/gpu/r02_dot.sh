set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02dot; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_levels.py -x -q -m gpu -k "dot" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for rep in 1 2; do for l in 8 9; do timeout -k 10 300 python tools/bench_kernels.py --level $l --only "dot" 2>&1 | grep "^dot" ; done; done
