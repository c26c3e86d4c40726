#!/bin/bash
# PMC passes of the apply kernel for two settings of an environment switch.  usage: r03_pmc_ab.sh <tag> <ENVVAR>
cd /tmp && export TMPDIR=/tmp
TAG=${1:-pmc}; VAR=${2:-HYTEG_HIP_APPLY_ALIGNED}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_$TAG; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
for v in 0 1; do
 for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum"; do
  N=$(echo $C | tr " " "_" | cut -c1-24)
  export $VAR=$v
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/v${v}_$N -- python3 $R/bench.py --steps 100 --warmup 20 --regions 2 --no-cpu-baseline > $O/v${v}_$N.log 2>&1 || echo "pass $v $N failed"
 done
done
python3 - <<PY
import csv, glob, collections
for v in (0, 1):
    for d in sorted(glob.glob("$O/v%d_*/" % v)):
        acc = collections.defaultdict(list)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "p1_apply_zmarch" in r["Kernel_Name"] or "calib_copy" in r["Kernel_Name"]:
                    acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, vals in sorted(acc.items()):
            print("$VAR=%d" % v, k[0], k[1], "mean", round(sum(vals) / len(vals), 1), "n", len(vals))
PY
