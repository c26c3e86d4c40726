set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02o; mkdir -p $O
for a in 0 1 2 3; do
echo "== ablation $a (1: no barrier, 2: no look-ahead sum)"
timeout -k 10 120 hyteg_amd/csrc/exp/sor_trace_a$a 8 > $O/sor_trace_l8_a$a.txt 2>&1 || { tail -20 $O/sor_trace_l8_a$a.txt; exit 1; }
grep "sweep\|T  24" $O/sor_trace_l8_a$a.txt
done
