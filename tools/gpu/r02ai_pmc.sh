set -e
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fp32.py tests/test_gpu_large_levels.py -m gpu -x -q -k "apply or jacobi or Jacobi or fp32 or level9 or largest" 2>&1 | tail -1
bash $R/tools/profile_bench.sh
ls $R/gpurun_out/pmc_bench | head
