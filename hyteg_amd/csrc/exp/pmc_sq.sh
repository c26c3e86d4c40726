#!/bin/bash
# usage: pmc_sq.sh <tag> <filter>   -- SQ / TCP / TCC counter passes for one harness kernel (separate passes, --pmc only)
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/hyteg_amd/csrc/exp/apply_bench
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
mkdir -p $O
i=0
while read -r C; do
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$1_$i -- $B 8 10 8 "$2" > $O/$1_$i.log 2>&1 || echo "set $i failed: $C"
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum
SETS
python3 - "$O" "$1" <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:40s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
