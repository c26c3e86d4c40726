#!/bin/bash
# HBM traffic of the P2 apply at level 7 from the PMC counters (separate passes for FETCH_SIZE and WRITE_SIZE), calibrated on the copy kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/r03_p2_pmc; mkdir -p $P
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $P/p2_$C -- python3 $R/tools/gpu/scratch/p2_pmc_driver.py > $P/p2_$C.log 2>&1 || echo "pmc pass $C failed"
  tail -1 $P/p2_$C.log
done
python3 - <<PY
import csv, glob, collections
P = "$P"
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{P}/p2_{c}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0][-70:]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if any(t in k for t in ("calib_copy", "p2_class_rows", "p2_apply_fused", "p2_rows_kernel")):
                res[k][c] = (sum(v) / len(v), len(v))
ne, nv = 2796160 * 7 // 7, 366145
import json
out = []
for k, d in res.items():
    out.append(f"{k}: " + ", ".join(f"{c} mean {m:.1f} over {n} launches" for c, (m, n) in d.items()))
open(f"{P}/p2_pmc_raw.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
