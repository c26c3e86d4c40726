set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02chain; mkdir -p $O
for f in 1 3; do
timeout -k 10 300 python3 hyteg_amd/csrc/exp/rank_kernel_probe.py $f > $O/probe_three_$f.txt 2>&1 || { tail -20 $O/probe_three_$f.txt; exit 1; }
grep "launches\|interior\|level" $O/probe_three_$f.txt
done
