cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02pmcmisc
mkdir -p $O
run_sets() { # tag, only-filter
  i=0
  while read -r C; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$1_s$i -- python3 $R/tools/bench_kernels.py --level 8 --reps 10 --only "$2" > $O/$1_s$i.log 2>&1 || echo "set $i failed: $C"
  done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
SETS
}
run_sets restrict "restrict (fine"
run_sets prolongate "prolongate Replace"
run_sets jacobi "Jacobi"
run_sets sor "SOR"
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for tag, pat in (("restrict","p1_restrict_kernel"),("prolongate","p1_prolongate_brick"),("jacobi","p1_apply_zmarch"),("sor","p1_sor_block")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{tag}_s*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(tag, k)
        for c, v in sorted(cs.items()):
            print(f"   {c:40s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
