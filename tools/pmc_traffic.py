#!/usr/bin/env python3
"""Post-process the PMC passes of tools/profile_bench.sh (gpurun_out/pmc_bench/) into profiles/pmc_traffic.json:
HBM bytes per launch of the apply kernel, FETCH_SIZE corrected with the factor calibrated in the SAME passes on bench.py's
copy-floor kernel (known size, same access width; MI355X_MICROARCH.md, HBM / rocprofv3 section).
Usage: python tools/pmc_traffic.py [gpurun_out/pmc_bench] [round tag]"""
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "pmc_bench"
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
COPY_BYTES = 2862209 * 8  # one level-8 cell array, read once and written once by the calibration copy kernel


def newest_per_pass(prefix):
    """gpurun merges every call's outputs into the same local tree: keep only the newest CSV of each pass directory"""
    out = []
    for d in sorted(src.glob(f"{prefix}_*")):
        files = sorted((q for q in d.rglob("*counter_collection.csv")), key=lambda q: q.stat().st_mtime)
        if files:
            out.append(str(files[-1]))
    return out


def mean_counter(prefix, kernel_substr, counter):
    vals = []
    for f in newest_per_pass(prefix):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


KERNEL = "p1_apply_zmarch_"  # p1_apply_zmarch_kernel or p1_apply_zmarch_preload_kernel (first arguments preloaded into SGPRs)
fetch, nf = mean_counter("bench", KERNEL, "FETCH_SIZE")
write, nw = mean_counter("bench", KERNEL, "WRITE_SIZE")
cfetch, _ = mean_counter("bench", "calib_copy_kernel<true>", "FETCH_SIZE")
cwrite, _ = mean_counter("bench", "calib_copy_kernel<true>", "WRITE_SIZE")
kname = None
for f in newest_per_pass("bench"):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            kname = r["Kernel_Name"].split("(")[0]
            break
    if kname:
        break
if None in (fetch, write, cfetch, cwrite):
    raise SystemExit(f"missing counters under {src}: fetch={fetch} write={write} calib fetch={cfetch} write={cwrite}")


def capi_kernel_name(rocprof_name):
    """'void hyteg_hip::p1_apply_zmarch_kernel<0, 4, 8, 0, false, 2, double>' -> the form hyteg_hip_p1_apply_kernel_name
    (and bench.py's roofline.kernel) uses: 'p1_apply_zmarch_kernel<MODE=0,NY=4,LZ=8,EX_AUX=0,DEC=0,PFD=2>'"""
    import re

    m = re.search(r"(p1_apply_zmarch_(?:preload_)?kernel)<([^>]*)>", rocprof_name)
    if not m:
        return None
    a = [x.strip() for x in m.group(2).split(",")]
    dec = {"false": "0", "true": "1"}.get(a[4], a[4])
    name = f"{m.group(1)}<MODE={a[0]},NY={a[1]},LZ={a[2]},EX_AUX={a[3]},DEC={dec},PFD={a[5]}>"
    return name if len(a) < 7 or a[6] == "double" else name[:-1] + f",T={a[6]}>"


import subprocess

try:
    git_head = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:  # noqa: BLE001
    git_head = None
kf, kw = COPY_BYTES / (cfetch * 1024), COPY_BYTES / (cwrite * 1024)
read_b, write_b = fetch * 1024 * kf, write * 1024 * kw
alg = 2731135 * 16
out = {
    "bytes_per_launch": int(round(read_b + write_b)),
    "kernel_name": capi_kernel_name(kname),
    "level": 8,
    "git_head": git_head,
    "read_bytes": int(round(read_b)),
    "write_bytes": int(round(write_b)),
    "kernel_as_rocprof_names_it": kname,
    "method": "rocprofv3 --kernel-trace --pmc, separate passes for FETCH_SIZE and WRITE_SIZE over `python3 bench.py --steps 100 "
              f"--warmup 20 --regions 2 --no-cpu-baseline` ({nf} / {nw} dispatches, level 8, rotating buffers); counters are in KiB "
              "(tools/profile_bench.sh, tools/pmc_traffic.py)",
    "raw": {"FETCH_SIZE_avg_KiB": fetch, "WRITE_SIZE_avg_KiB": write, "calib_FETCH_SIZE_avg_KiB": cfetch, "calib_WRITE_SIZE_avg_KiB": cwrite},
    "gfx950_correction": f"bench.py's copy-floor kernel in the same passes (8 B per lane, {COPY_BYTES} B known each way): FETCH_SIZE*1024*{kf:.3f} = known bytes, "
                         f"WRITE_SIZE*1024*{kw:.3f} = known bytes; the apply's figures use these factors "
                         "(FETCH_SIZE reports half of a coalesced stream on gfx950, MI355X_MICROARCH.md)",
    "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": round((read_b + write_b) / alg, 3),
    "round": tag,
}
(ROOT / "profiles" / "pmc_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
