set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02c; mkdir -p $O
cd hyteg_amd/csrc/exp
./apply_trace_time 8 400 9 > $O/trace_time_l8.txt 2>&1
./apply_trace_time 8 400 9 > $O/trace_time_l8_again.txt 2>&1
./apply_trace 8 100 9 > $O/trace_l8.txt 2>&1
cat $O/trace_time_l8.txt
