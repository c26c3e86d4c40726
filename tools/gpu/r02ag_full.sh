set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02ag; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || { tail -60 $O/pytest_gpu.txt; exit 1; }
tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
