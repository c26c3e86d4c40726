set -e
for f in 1 3; do
for d in 16 17; do
echo "== faces $f dbg $d"
HYTEG_HIP_RANK_DBG=$d timeout -k 10 300 python hyteg_amd/csrc/exp/rank_kernel_probe.py $f 2>&1 | grep "launches"
done
done
