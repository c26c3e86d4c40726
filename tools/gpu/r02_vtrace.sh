set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02vtrace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/vcycle_breakdown.py --smoother jacobi --max 8 --cycles 10 > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]: print(r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
PY
