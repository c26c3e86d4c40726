set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02p2xcd; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_p2.py tests/test_gpu_p2_gmg.py tests/test_gpu_p2_config4.py -x -q -m gpu > $O/pytest_p2.txt 2>&1 || { tail -40 $O/pytest_p2.txt; exit 1; }
tail -2 $O/pytest_p2.txt
timeout -k 10 300 python tools/bench_p2_apply.py > $O/p2_apply_dpp.txt 2>&1 || { tail -20 $O/p2_apply_dpp.txt; exit 1; }
grep "level 7\|level 6 P2 apply" $O/p2_apply_dpp.txt
echo "---- HYTEG_HIP_P2_ROWS_DPP=0"
HYTEG_HIP_P2_ROWS_DPP=0 timeout -k 10 300 python tools/bench_p2_apply.py --levels 6 7 > $O/p2_apply_nodpp.txt 2>&1
grep "level 7\|level 6 P2 apply" $O/p2_apply_nodpp.txt
