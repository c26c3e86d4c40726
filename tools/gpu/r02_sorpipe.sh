set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02sorpipe; mkdir -p $O
timeout -k 10 900 python tools/bench_kernels.py --reps 20 > $O/table.txt 2>&1 || { tail -30 $O/table.txt; exit 1; }
grep "V(3,3)\|Stokes" $O/table.txt
HYTEG_HIP_SOR_PIPELINE=0 timeout -k 10 900 python tools/bench_kernels.py --reps 20 > $O/table_nopipe.txt 2>&1
echo "--- HYTEG_HIP_SOR_PIPELINE=0"; grep "V(3,3) GS" $O/table_nopipe.txt
