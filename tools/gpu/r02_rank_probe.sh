set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02chain; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in 1 3; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$f -- python3 $R/hyteg_amd/csrc/exp/rank_kernel_probe.py $f > $O/probe$f.txt 2>&1
grep "launches\|interior" $O/probe$f.txt
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof$f/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print('   ', r['Name'][:80], r['Calls'], r['AverageNs'])
PY
done
