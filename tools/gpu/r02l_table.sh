set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02l; mkdir -p $O
python tools/bench_kernels.py > $O/kernel_table.txt 2>&1 || { tail -30 $O/kernel_table.txt; exit 1; }
grep -v "^{" $O/kernel_table.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py --reps 50 > $O/prof.log 2>&1 || echo "rocprof pass failed"
find $O/prof -name "*kernel_stats.csv" | head -2
