#!/bin/bash
# What kind of box is this?  Boxes of the pool run the copy floor within 10 % of each other but bench.py's apply between 9.1
# and 12.8 us -- and on one box bench.py measured 12.8 us where tools/apply_shape_sweep.py measured 9.3 us for the same kernel
# in the same call.  rocm-smi, shader clock under load, allocation probe, bench.py (5 regions), level-8 shape sweep.
# usage: r03_box_probe.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_box_$1; mkdir -p $O
{ rocm-smi --showclocks --showpower --showtemp --showmeminfo vram --showcomputepartition --showmemorypartition 2>&1 | grep -v "^$" | head -60; rocminfo 2>/dev/null | grep -i "Compute Unit\|Max Clock\|Marketing\|Memory Properties\|Size:" | head -30; } > $O/smi_before.txt
timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --regions 5 > $O/bench.json 2> $O/bench.err || echo "bench failed"
timeout -k 10 300 python3 $R/tools/gpu/scratch/alloc_probe.py > $O/alloc_probe.txt 2>&1 || echo "alloc probe failed"
timeout -k 10 300 python3 $R/tools/apply_shape_sweep.py --levels 8 --rounds 3 > $O/sweep.txt 2>&1 || echo "sweep failed"
timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --regions 5 > $O/bench2.json 2> $O/bench2.err || echo "bench failed"
for f in bench bench2; do tail -1 $O/$f.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f launch_us', round(r['launch_us'],3), 'frac', round(r['frac'],4), 'copy_us', round(r['copy_us'],3), 'frac_of_copy', round(r['frac_of_copy'],3))"; done
grep -i "partition" $O/smi_before.txt | head -4
grep -v amdgpu.ids $O/alloc_probe.txt
grep -i "apply Replace" $O/sweep.txt | head -8
