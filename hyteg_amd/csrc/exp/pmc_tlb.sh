#!/bin/bash
# usage: pmc_tlb.sh <tag> <level> <filter>
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/hyteg_amd/csrc/exp/apply_bench
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_tlb
mkdir -p $O
i=0
while read -r C; do
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$1_$i -- $B $2 10 8 "$3" > $O/$1_$i.log 2>&1 || echo "set $i failed: $C"
done <<'SETS'
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
FETCH_SIZE
WRITE_SIZE
SETS
python3 - "$O" "$1" <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:40s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
