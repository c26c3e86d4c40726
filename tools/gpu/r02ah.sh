set -e
for rep in 1 2 3 4; do
for v in 0 1; do
echo -n "preload $v: "; HYTEG_HIP_APPLY_PRELOAD=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e9,1), round(r['launch_us'],2), round(r['frac'],3))"
done
done
