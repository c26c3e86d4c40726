set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02b; mkdir -p $O
cd hyteg_amd/csrc/exp
./icache_probe > $O/icache_probe.txt 2>&1
./apply_trace_time 8 300 9 > $O/trace_time_l8.txt 2>&1
./apply_trace_time 7 300 9 > $O/trace_time_l7.txt 2>&1
./apply_trace_time 9 50 3 > $O/trace_time_l9.txt 2>&1
./apply_trace 8 100 9 > $O/trace_l8.txt 2>&1
cat $O/icache_probe.txt $O/trace_time_l8.txt
