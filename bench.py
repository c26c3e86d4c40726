#!/usr/bin/env python3
"""bench.py -- headline benchmark: P1 Laplace apply() on level-8 macro-cells, DoF-updates/s + HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--level L] [--no-cpu-baseline]

A "step" is ONE P1ConstantLaplaceOperator::apply(src, dst, level, Inner, Replace) over the rank's macro-cell
(one kernel launch through the C-ABI).  Inputs are resident in HBM before the timed region; a ring of
buffer pairs larger than the 256 MiB Infinity Cache is cycled so that every launch streams from HBM.
For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank owns one macro-cell
(weak scaling); rank 0 prints ONE JSON line with the whole-job aggregate.

value      = (interior DoFs per cell) * (cells) * K / (max over ranks of the timed region)
roofline   = algorithmic bytes per launch (16 B per DoF-update) / average launch duration
             measured with HIP events on the launch stream, against the 8 TB/s HBM3E peak.
cpu_baseline = the CPU restatement of the reference kernel (oracle/, -O3 -march=native, 1 thread) timed on
             this box's host cores on rank 0 at N=1, on a bounded number of level-8 sweeps.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MALL_BYTES = 256 * 1024 * 1024
REF_TET = ((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))


def laplace_stencil(level: int):
    """15 Laplace weights of the unit reference tet at `level`.  The level-2 weights are exact rationals
    (multiples of h/3 with h = 1/4); HyTeG's weights scale by 1/2 per level
    (tests/hyteg/vertexdofspace/VertexDoFStencilAssemblyTest.cpp:85).  Order: include/hyteg_hip.h."""
    h = 0.25
    t = h / 3.0
    lvl2 = [-4 * t, t, t, -t, -2 * t, -t, -4 * t, 20 * t, -4 * t, -t, -2 * t, -t, t, t, -4 * t]
    return [v * 0.5 ** (level - 2) for v in lvl2]


def cpu_baseline(level: int, w, budget_s: float = 12.0):
    """Time the CPU restatement of apply_3D_macrocell_vertexdof_to_vertexdof_replace (oracle/, test
    infrastructure used here ONLY as the reported baseline), 1 thread, protocol of
    apps/benchmarks/KernelBench/3DKernelBench.cpp:59-82 (double the sweeps until the block is long enough)."""
    import numpy as np

    from oracle import p1_oracle as po

    n = po.cell_size(level)
    rng = np.random.default_rng(42)
    src = rng.random(n)
    dst = np.zeros(n)
    for _ in range(2):
        po.apply_cell(dst, src, level, w, fast=True)
    sweeps, total_t, total_sweeps = 1, 0.0, 0
    while total_t < budget_s:
        t0 = time.perf_counter()
        for _ in range(sweeps):
            po.apply_cell(dst, src, level, w, fast=True)
        dt = time.perf_counter() - t0
        total_t += dt
        total_sweeps += sweeps
        last = dt / sweeps
        if dt > 0.5 * budget_s:
            break
        sweeps *= 2
    inner = po.cell_inner_size(level)
    return {
        "value": inner / last,
        "unit": "DoF-updates/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{total_sweeps} apply sweeps of one level-{level} macro-cell ({inner} DoF-updates each), "
                  f"{total_t:.1f} s, gcc -O3 -march=native, last block {last * 1e3:.2f} ms/sweep",
        "ms_per_sweep": last * 1e3,
        "host_cpus": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--level", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    from hyteg_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    level = args.level
    capi.lib()
    capi.prepare_level(level)
    n = capi.cell_size(level)
    inner = capi.cell_inner_size(level)
    w = laplace_stencil(level)

    # ring of buffer pairs > Infinity Cache so that each launch reads and writes HBM
    pair_bytes = 2 * n * 8
    nbuf = max(2, -(-int(1.5 * MALL_BYTES) // pair_bytes))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42 + rank)
    srcs = [torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) for _ in range(nbuf)]
    dsts = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
    sp = [t.data_ptr() for t in srcs]
    dp = [t.data_ptr() for t in dsts]
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream

    def step(k):
        capi.p1_apply_cell(dp[k % nbuf], sp[k % nbuf], level, w, capi.REPLACE, sh)

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for k in range(args.steps):
        step(k)
    ev1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the launch stream

    if dist is not None:
        tmax = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = float(tmax[0]), float(tmax[1])

    if rank == 0:
        launch_us = dev_ms * 1e3 / args.steps
        algo_bytes = 16 * inner  # 8 B compulsory src read + 8 B dst write per DoF-update
        achieved = algo_bytes / (launch_us * 1e-6) / 1e9
        traffic = None
        tfile = ROOT / "profiles" / "pmc_traffic.json"
        if tfile.exists():
            try:
                traffic = json.loads(tfile.read_text()).get("p1_apply_tiled_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "DoF-updates/s, P1 Laplace apply() level 8",
            "value": inner * world * args.steps / elapsed,
            "unit": "DoF-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"P1ConstantLaplaceOperator::apply(Replace), one level-{level} macro-cell per GPU "
                            f"(unit reference tet, {n} entries, {inner} DoF-updates per apply), "
                            f"{nbuf} rotating buffer pairs ({nbuf * pair_bytes / 2**20:.0f} MiB > 256 MiB Infinity Cache)",
                "level": level,
                "macro_cells": world,
                "halo_exchange": False if world > 1 else None,
                "device": capi.device_name(),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "p1_apply_zmarch_kernel<REPLACE,4,4>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "launch_us": launch_us,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(level, w)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
