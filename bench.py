#!/usr/bin/env python3
"""bench.py -- headline benchmark: P1 Laplace apply() on level-8 macro-cells, DoF-updates/s + HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--level L] [--no-cpu-baseline]

A "step" is ONE P1ConstantLaplaceOperator::apply(src, dst, level, Inner, Replace) through the C++ host layer
(hyteg_amd/host, the mirror of HyTeG's operator API), which calls the HIP kernels through the C-ABI.  Inputs are
resident in HBM before the timed region; a ring of function pairs whose SOURCE arrays alone are 2.2 x the 256 MiB
Infinity Cache is cycled so that every apply streams from HBM (a ring that exceeds the cache only in total is read
from it: the destination's nontemporal stores do not occupy the cache -- see the comment at the ring below).  For N > 1 (launched by torch.distributed.run, one rank per GPU) the mesh has
N macro-cells, one per GPU (weak scaling); the shares of the macro-face/edge/vertex DoFs the cells have in common
are exchanged over RCCL inside every apply, overlapped with the interior kernel.  Rank 0 prints ONE JSON line with
the whole-job aggregate.

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


def _time_sweeps(fn, budget_s):
    """protocol of apps/benchmarks/KernelBench/3DKernelBench.cpp:59-82: double the sweeps until the block is long enough"""
    sweeps, total_t, total_sweeps = 1, 0.0, 0
    while True:
        t0 = time.perf_counter()
        for _ in range(sweeps):
            fn()
        dt = time.perf_counter() - t0
        total_t += dt
        total_sweeps += sweeps
        last = dt / sweeps
        if dt > 0.5 * budget_s or total_t >= budget_s:
            return last, total_sweeps, total_t
        sweeps *= 2


def cpu_baseline(level: int, w, budget_s: float = 10.0, threads: int = 16):
    """Time the CPU restatement of apply_3D_macrocell_vertexdof_to_vertexdof_replace (oracle/, test infrastructure used
    here ONLY as the reported baseline): (a) 1 thread, as the reference kernel runs per MPI rank, (b) `threads` threads,
    one macro-cell per thread (emulates that many ranks of the reference on this host)."""
    import threading

    import numpy as np

    from oracle import p1_oracle as po

    n = po.cell_size(level)
    inner = po.cell_inner_size(level)
    rng = np.random.default_rng(42)
    src = rng.random(n)
    dst = np.zeros(n)
    for _ in range(2):
        po.apply_cell(dst, src, level, w, fast=True)
    last, total_sweeps, total_t = _time_sweeps(lambda: po.apply_cell(dst, src, level, w, fast=True), budget_s)

    # one macro-cell per thread (ctypes releases the GIL inside the C sweep); every thread sweeps its own arrays
    threads = max(1, min(threads, os.cpu_count() or 1))
    srcs = [rng.random(n) for _ in range(threads)]
    dsts = [np.zeros(n) for _ in range(threads)]
    reps = max(2, int(0.5 * budget_s / last))
    barrier = threading.Barrier(threads + 1)

    def work(k):
        po.apply_cell(dsts[k], srcs[k], level, w, fast=True)
        barrier.wait()
        for _ in range(reps):
            po.apply_cell(dsts[k], srcs[k], level, w, fast=True)
        barrier.wait()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    for t in ts:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    barrier.wait()
    dt_multi = time.perf_counter() - t0
    for t in ts:
        t.join()
    return {
        "value": inner / last,
        "unit": "DoF-updates/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{total_sweeps} apply sweeps of one level-{level} macro-cell ({inner} DoF-updates each), "
                  f"{total_t:.1f} s, gcc -O3 -march=native, last block {last * 1e3:.2f} ms/sweep",
        "ms_per_sweep": last * 1e3,
        "host_cpus": os.cpu_count(),
        "one_cell_per_thread": {
            "value": inner * reps * threads / dt_multi,
            "unit": "DoF-updates/s",
            "cores": threads,
            "sample": f"{threads} threads x {reps} sweeps, one level-{level} macro-cell per thread, {dt_multi:.1f} s",
        },
    }


MESH_FOR_WORLD = {1: "tet_1el", 2: "pyramid_2el", 4: "pyramid_4el", 8: "regular_octahedron_8el"}


def level9_supporting(host, capi, storage, stream, region, rng):
    """P1ConstantLaplaceOperator::apply at level 9 on the same macro-cell: K steps per region like the headline, ring of
    pairs whose sources exceed the Infinity Cache 2.2 times, median of 5 regions"""
    lv = 9
    n9 = capi.cell_size(lv)
    pair_bytes = 2 * n9 * 8 * storage.n_local_cells
    nb = max(2, -(-int(4.4 * MALL_BYTES) // pair_bytes))
    op = host.P1ConstantOperator(storage, lv, lv)
    ss = [host.P1Function(storage, f"s9_{k}", lv, lv) for k in range(nb)]
    ds = [host.P1Function(storage, f"d9_{k}", lv, lv) for k in range(nb)]
    try:
        for f in ss:
            for c in range(storage.n_local_cells):
                f.upload_cell(c, lv, rng.random(n9))
            f.sync_shared(lv, host.All)
        steps9 = op.prepared_cycle(ss, ds, lv, host.Inner, host.Replace)
        steps9(0, 4 * nb)
        devs = sorted(region(steps9)[1] for _ in range(5))
        us = devs[len(devs) // 2] * 1e3 / region.steps
        algo = 16 * capi.cell_inner_size(lv)
        return {"level": lv, "launch_us": us, "algorithmic_bytes_per_launch": algo, "achieved": algo / (us * 1e-6) / 1e9,
                "frac": algo / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "ring_pairs": nb, "regions": 5,
                "kernel": capi.p1_apply_kernel_name(lv, capi.REPLACE),
                "note": "same kernel template and measurement as the headline, one level finer (8 x the DoFs per launch); supporting "
                        "evidence for where the level-8 launch loses its time, not BASELINE.json's configuration"}
    finally:
        for o in ss + ds + [op]:
            o.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--level", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--regions", type=int, default=25,
                    help="the timed K-step region is repeated this many times (each one bracketed by barrier + synchronize); "
                         "value / ms_per_step / launch_us are the MEDIAN region, min and first region are reported beside it")
    args = ap.parse_args()

    import numpy as np
    import torch

    from hyteg_amd import capi, host

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if world not in MESH_FOR_WORLD:
        raise SystemExit(f"bench.py supports 1, 2, 4 or 8 GPUs (one macro-cell per GPU), not {world}")
    # rehearsal on a box with fewer GPUs than ranks (development only): HYTEG_BENCH_BACKEND=gloo HYTEG_BENCH_SHARE_GPU=1
    backend = os.environ.get("HYTEG_BENCH_BACKEND", "nccl")
    if os.environ.get("HYTEG_BENCH_SHARE_GPU") == "1":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    # N > 1: before this process touches its GPU, a child process per rank tries the peer-to-peer exchange on its own
    # (hyteg_amd/p2p_canary.py: IPC mappings between the ranks' GPUs, the real pack / wait kernels, values checked) -- if the
    # GPUs cannot reach each other's memory that way, it is the canary that fails, not the benchmark
    canary = None
    try_p2p = world > 1 and os.environ.get("HYTEG_BENCH_P2P", "0") == "1"
    if try_p2p:
        from hyteg_amd import p2p_canary

        canary = p2p_canary.launch(rank, world, local_rank, f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    level = args.level
    capi.lib()
    host.lib()
    capi.prepare_level(level)
    n = capi.cell_size(level)

    # one level-`level` macro-cell per GPU: meshes of 1, 2, 4, 8 tetrahedra from the reference's test data; the cells of
    # a rank-r process are those with (cell id % world) == r (HyTeG's round-robin default)
    mesh = ROOT / "hyteg_amd" / "data" / "meshes" / f"{MESH_FOR_WORLD[world]}.msh"
    if world == 1 and os.environ.get("HYTEG_BENCH_MESH"):
        # development only: several macro-cells on ONE rank (boundary shares and the local reduce without any exchange)
        mesh = ROOT / "hyteg_amd" / "data" / "meshes" / f"{os.environ['HYTEG_BENCH_MESH']}.msh"
    storage = host.Storage.from_gmsh(mesh, rank, world)
    stream = torch.cuda.current_stream()
    storage.set_stream(stream.cuda_stream)
    ctx = None
    if world > 1:
        from hyteg_amd.distributed import DistributedContext

        ctx = DistributedContext(storage, [level], torch.device("cuda", local_rank))
    laplace = host.P1ConstantOperator(storage, level, level)  # assembles the cell / face / edge / vertex stencils

    # Ring of function pairs so that each apply reads and writes HBM.  The SOURCE arrays alone must exceed the 256 MiB Infinity
    # Cache more than twice over: the destination is written with nontemporal stores, which do not stay in that cache, so a ring
    # whose pairs together exceed it (rounds 1-2 and most of round 3: 9 pairs = 393 MiB, sources 197 MiB) still served every
    # READ from the Infinity Cache -- 9.2 us per launch where a ring of 12 and more pairs measures 13.2-13.9 (round 3,
    # tools/gpu/scratch/alloc_probe.py).  The small ring is measured once more below and reported beside the
    # headline as roofline.infinity_cache_assisted.
    pair_bytes = 2 * n * 8 * storage.n_local_cells
    nbuf = max(2, -(-int(4.4 * MALL_BYTES) // pair_bytes))
    nbuf_small = max(2, -(-int(1.5 * MALL_BYTES) // pair_bytes))
    rng = np.random.default_rng(42 + rank)
    srcs = [host.P1Function(storage, f"src{k}", level, level) for k in range(nbuf)]
    dsts = [host.P1Function(storage, f"dst{k}", level, level) for k in range(nbuf)]
    for f in srcs:
        for c in range(storage.n_local_cells):
            f.upload_cell(c, level, rng.random(n))
        f.sync_shared(level, host.All)  # copies of a shared DoF hold one value
    one = host.P1Function(storage, "one", level, level)
    one.interpolate(1.0, level, host.All)
    inner_dofs = int(round(one.dot(one, level, host.Inner)))  # global number of DoFs an apply(..., Inner) updates
    one.close()

    def step(k):
        # P1ConstantLaplaceOperator::apply( src, dst, level, Inner, Replace ): boundary shares -> halo exchange started
        # -> interior stencil kernel (overlaps the exchange) -> reduction of the shares
        laplace.apply(srcs[k % nbuf], dsts[k % nbuf], level, host.Inner, host.Replace)

    def all_ranks(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def results(pairs):
        """apply on the given ring pairs -> the destination arrays of this rank's cells (bytes are compared)"""
        out = []
        for k in pairs:
            step(k)
            out += [dsts[k % nbuf].download_cell(c, level) for c in range(storage.n_local_cells)]
        return out

    # N > 1: the exchange goes peer to peer (pack kernels store into the neighbour GPUs' IPC-mapped arenas, comm_p2p.hip)
    # if -- and only if -- that reproduces, bit for bit on every rank, what the RCCL send/recv transport underneath delivers,
    # before and after the timed region; otherwise the run is (re)done on RCCL.  HYTEG_BENCH_P2P=0 skips the attempt.
    p2p = {"tried": False}
    check_pairs = [0, 1, 0, 1 % nbuf, 0]  # both slot parities, repeated
    if try_p2p:
        ok, text = p2p_canary.finish(canary)
        if not all_ranks(ok):
            try_p2p = False
            ctx.transport_note = f"peer-to-peer canary failed on some rank (this rank: {'ok' if ok else text.strip()[-200:]}): run on {ctx.transport}"
    if try_p2p:
        reference = results(check_pairs)
        p2p["tried"] = True
        if ctx.enable_p2p():
            ok, err = True, ""
            try:
                got = results(check_pairs)
                storage.check_transport()
                ok = all(np.array_equal(a, b) for a, b in zip(got, reference))
            except Exception as e:  # noqa: BLE001 -- agreed on below
                ok, err = False, repr(e)
            if not all_ranks(ok):
                ctx.disable_p2p(f"peer-to-peer exchange did not reproduce the RCCL results before the timed region {err}: run on "
                                f"{ctx.inner_transport}")
        p2p["used_before"] = ctx.transport == "p2p"

    # `count` applies, step k on ring pair (first + k) % nbuf, issued by the C++ host layer's own loop (what a C++
    # application writes around apply(); one ctypes call for the whole region, its handle arrays built here)
    apply_steps = laplace.prepared_cycle(srcs, dsts, level, host.Inner, host.Replace)
    apply_steps_small_ring = laplace.prepared_cycle(srcs[:nbuf_small], dsts[:nbuf_small], level, host.Inner, host.Replace)

    copy_ptrs = [[(dsts[k].cell_pointer(c, level), srcs[k].cell_pointer(c, level)) for c in range(storage.n_local_cells)]
                 for k in range(nbuf)]

    # the copy floor: the same ring pairs, every cell array read once and written once (nontemporal stores), same stream,
    # issued by one C loop like the applies
    if storage.n_local_cells == 1:
        copy_steps = capi.calib_copy_ring([p[0] for p in copy_ptrs], n, True, stream.cuda_stream)
    else:
        def copy_steps(first, count, ev_start=None, ev_stop=None):
            if ev_start:
                capi.event_record(ev_start, stream.cuda_stream)
            for k in range(first, first + count):
                for d, s_ in copy_ptrs[k % nbuf]:
                    capi.calib_copy(d, s_, n, True, stream.cuda_stream)
            if ev_stop:
                capi.event_record(ev_stop, stream.cuda_stream)

    ev0, ev1 = capi.event_create_timing(), capi.event_create_timing()

    def region(fn):
        """EXACTLY K steps between barrier + synchronize on both sides (wall clock) and between two HIP events recorded on
        the launch stream by the C loop that issues the K steps, directly before the first and after the last launch (device
        time; an event recorded from Python sits 5-8 us of interpreter time in front of the first launch, which a K = 20
        region of an otherwise idle GPU sees as 0.3 us per step: profiles/r03_k20_probe.txt)"""
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(0, args.steps, ev0, ev1)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, capi.event_elapsed_ms(ev0, ev1)

    region.steps = args.steps

    def measure():
        # untimed: every ring pair is touched (twice) whatever --warmup says, so that no first access (page tables, TLB) of a
        # buffer falls into the timed region; then the W warm-up steps of the contract
        apply_steps(0, pre_warm)
        apply_steps(0, args.warmup)
        # R regions of K steps each.  The copy floor (N = 1) is measured by R more regions AFTER the apply regions, not between
        # them: a copy region in front of an apply region changes what the Infinity Cache holds and made short (K = 20) apply
        # regions 19 % slower (profiles/r03_region_probe.txt)
        walls, devs, copies = [], [], []
        for r in range(max(1, args.regions)):
            w_, d_ = region(apply_steps)
            walls.append(w_)
            devs.append(d_)
        if world == 1 and os.environ.get("HYTEG_BENCH_COPY", "1") != "0":
            copy_steps(0, nbuf)
            for r in range(max(1, args.regions)):
                copies.append(region(copy_steps)[1])
        # the ring of rounds 1-2 (sources fit into the Infinity Cache), for continuity: NOT the headline
        small.clear()
        if world == 1 and os.environ.get("HYTEG_BENCH_SMALL_RING", "1") != "0":
            apply_steps_small_ring(0, 4 * nbuf_small)
            for r in range(min(9, max(1, args.regions))):
                small.append(region(apply_steps_small_ring)[1])

        # the same K steps once more, every step between its own pair of events (reported beside the region mean: an upper
        # bound, an event between two kernels costs time of its own)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        evs[0].record(stream)
        for k in range(args.steps):
            step(k)
            evs[k + 1].record(stream)
        torch.cuda.synchronize()
        per_step_us = sorted(evs[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(args.steps))
        return walls, devs, copies, per_step_us

    # untimed, before the W warm-up steps: the GPU's clocks follow the load over tens of milliseconds, and a run of 25 regions of
    # K = 20 steps is 5 ms of work after seconds of set-up -- without this its regions measure the ramp (10.1-10.3 us per launch
    # against 9.5 with it and 9.3 for K = 2000, same box; HYTEG_BENCH_PREWARM=0 switches it off, at least 2 * nbuf applies remain)
    small = []
    pre_warm = max(2 * nbuf, int(os.environ.get("HYTEG_BENCH_PREWARM", "10000")))
    walls, devs, copies, per_step_us = measure()
    if world > 1 and ctx.transport == "p2p":
        ok, err = True, ""
        try:
            got = results(check_pairs)
            storage.check_transport()
            ok = all(np.array_equal(a, b) for a, b in zip(got, reference))
        except Exception as e:  # noqa: BLE001
            ok, err = False, repr(e)
        if not all_ranks(ok):
            ctx.disable_p2p(f"peer-to-peer exchange failed its check after the timed region {err}: measured again on {ctx.inner_transport}")
            walls, devs, copies, per_step_us = measure()
    p2p["verified_on_check_pairs"] = world > 1 and ctx.transport == "p2p"
    median_us = per_step_us[len(per_step_us) // 2]

    if dist is not None:
        # MAX over ranks, region by region
        tmax = torch.tensor(walls + devs + [median_us], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        R_ = len(walls)
        walls, devs, median_us = [float(x) for x in tmax[:R_]], [float(x) for x in tmax[R_:2 * R_]], float(tmax[2 * R_])

    def med(v):
        v = sorted(v)
        return v[len(v) // 2]

    elapsed, dev_ms = med(walls), med(devs)

    if rank == 0:
        w = laplace.stencils(0, level)[0]
        launch_us = dev_ms * 1e3 / args.steps  # one event pair around a whole timed region / K, median region
        copy_us = (med(copies) * 1e3 / args.steps / storage.n_local_cells) if copies else None
        cell_inner = capi.cell_inner_size(level)
        algo_bytes = 16 * cell_inner  # 8 B compulsory src read + 8 B dst write per DoF-update, one cell
        achieved = algo_bytes / (launch_us * 1e-6) / 1e9
        kernel_name = capi.p1_apply_kernel_name(level, capi.REPLACE)
        # HBM traffic of that kernel from the PMC passes of tools/profile_bench.sh: reported only if the recorded file
        # is for the instantiation this run launched
        traffic, traffic_note = None, "no profiles/pmc_traffic.json"
        tfile = ROOT / "profiles" / "pmc_traffic.json"
        if tfile.exists():
            try:
                rec = json.loads(tfile.read_text())
                if rec.get("kernel_name") == kernel_name and rec.get("level") == level:
                    traffic = rec.get("bytes_per_launch")
                    traffic_note = f"profiles/pmc_traffic.json (git {rec.get('git_head')}, {rec.get('round')})"
                else:
                    traffic_note = f"profiles/pmc_traffic.json is for {rec.get('kernel_name')!r} at level {rec.get('level')}: not this kernel"
            except Exception as e:  # noqa: BLE001
                traffic_note = f"profiles/pmc_traffic.json unreadable: {e}"
        out = {
            "metric": "DoF-updates/s, P1 Laplace apply() level 8",
            "value": inner_dofs * args.steps / elapsed,
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
                "workload": f"P1ConstantLaplaceOperator::apply(src, dst, level {level}, Inner, Replace) through the host layer, "
                            f"one level-{level} macro-cell per GPU ({MESH_FOR_WORLD[world]}.msh, {n} entries per cell array, "
                            f"{inner_dofs} inner DoF-updates per apply over all GPUs), "
                            f"{nbuf} rotating function pairs ({nbuf * pair_bytes / 2**20:.0f} MiB per GPU; the source arrays alone are "
                            f"{nbuf * pair_bytes / 2 / MALL_BYTES:.1f} x the 256 MiB Infinity Cache)",
                "level": level,
                "macro_cells": world,
                "pre_warm_applies": pre_warm,
                "pre_warm_note": "untimed applies BEFORE the W warm-up steps of the contract: two passes over the buffer ring (no first access "
                                 "inside the timed region) and, by default, 10000 applies (~0.1 s) that bring the GPU's clocks up; "
                                 "HYTEG_BENCH_PREWARM=0 leaves only the ring passes (K = 20: 10.1-10.3 us per launch instead of 9.5, DESIGN 3.1)",
                "mesh_note": "1/2/4/8 GPUs run tet_1el / pyramid_2el / pyramid_4el / regular_octahedron_8el (one macro-cell per GPU);"
                             " they stand in for the MultigridStudies cube, whose 6 or 24 cells do not give one cell per GPU",
                "halo_exchange": (f"transport '{ctx.transport}': "
                                  + ("ncclSend/ncclRecv groups issued by the C++ host layer on a communication stream (RCCL over xGMI), "
                                     "event-ordered, overlapped with the interior kernel" if ctx.transport == "rccl" else
                                     f"pack kernel stores into the neighbour GPUs' IPC-mapped {ctx.p2p_arena['kind']} arenas over xGMI, "
                                     "the reduce kernel waits for their sequence numbers, no library call per exchange; canary process passed, results equal to the "
                                     f"'{ctx.inner_transport}' transport's bit for bit on every rank before and after the timed region"
                                     if ctx.transport == "p2p" else
                                     "torch.distributed all_to_all hooks (rehearsal transport)")
                                  + (f" [{ctx.transport_note}]" if ctx.transport_note else "")
                                  if world > 1 else None),
                "device": capi.device_name(),
                "device_properties": (lambda q: {"compute_units": q.multi_processor_count, "memory_GiB": round(q.total_memory / 2**30, 1),
                                                 "clock_rate_MHz": getattr(q, "clock_rate", 0) / 1e3 or None})(
                                                     torch.cuda.get_device_properties(torch.cuda.current_device())),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": achieved if world == 1 else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if world == 1 else None,
                "traffic": traffic,
                "traffic_source": traffic_note,
                "launch_us": launch_us,
                "launch_us_min_region": min(devs) * 1e3 / args.steps,
                "launch_us_first_region": devs[0] * 1e3 / args.steps,
                "frac_min_region": (algo_bytes / (min(devs) * 1e-3 / args.steps) / 1e9 / HBM_PEAK_GBS) if world == 1 else None,
                "copy_us": copy_us,
                "copy_us_min_region": (min(copies) * 1e3 / args.steps / storage.n_local_cells) if copies else None,
                "copy_bytes_per_launch": 16 * n if copies else None,
                "frac_of_copy": (copy_us / launch_us) if copy_us else None,
                "per_step_event_median_us": median_us,
                "per_step_event_min_us": per_step_us[0],
                "algorithmic_bytes_per_launch": algo_bytes,
                "infinity_cache_assisted": ({
                    "launch_us": sorted(small)[len(small) // 2] * 1e3 / args.steps,
                    "frac": algo_bytes / (sorted(small)[len(small) // 2] * 1e-3 / args.steps) / 1e9 / HBM_PEAK_GBS,
                    "ring_pairs": nbuf_small,
                    "regions": len(small),
                    "note": "the same K steps over the first ring_pairs pairs only -- the ring of rounds 1-2, whose source arrays "
                            "together fit into the 256 MiB Infinity Cache: its reads do not come from HBM, so this is NOT a fraction "
                            "of the HBM roofline; kept for comparison with the earlier rounds' lines"} if small else None),
                "note": "achieved = algorithmic bytes of one cell's interior kernel / launch_us; launch_us = time between two HIP"
                        " events recorded on the launch stream around the K applies of a wall-clock region / K, MEDIAN over"
                        " `regions` regions (N=1: one kernel per apply; every ring pair was touched before the first region)."
                        " copy_us: the same measurement (same ring pairs, same stream, same events, `regions` regions of K copies"
                        " after the apply regions) for hyteg_hip_calib_copy, the fastest copy kernel we know for this size"
                        " (plain loads, nontemporal stores), which reads and writes every entry of the cell array once"
                        " (copy_bytes_per_launch); frac_of_copy = copy_us / launch_us."
                        " per_step_event_median_us: the K steps once more with an event pair around EACH apply -- an upper"
                        " bound (the events cost time between the kernels), not used for the roofline figure",
            },
            "regions": {
                "count": len(walls),
                "reported": "median region (value, ms_per_step, roofline.launch_us); each region = exactly K steps between barrier + synchronize",
                "ms_per_step_median": elapsed * 1e3 / args.steps,
                "ms_per_step_min": min(walls) * 1e3 / args.steps,
                "ms_per_step_first": walls[0] * 1e3 / args.steps,
                "value_first_region": inner_dofs * args.steps / walls[0],
                "value_best_region": inner_dofs * args.steps / min(walls),
                "launch_us_all": [round(d * 1e3 / args.steps, 3) for d in devs],
                "copy_us_all": [round(d * 1e3 / args.steps / storage.n_local_cells, 3) for d in copies],
            },
        }
        if world == 1 and level == 8 and os.environ.get("HYTEG_BENCH_LEVEL9", "1") != "0":
            # supporting figure, NOT the headline: the same kernel one refinement level up, where one launch lasts long enough
            # for the fixed cost per launch (start and tail of the grid, ~3 us) to stop mattering
            try:
                out["roofline"]["level9_supporting"] = level9_supporting(host, capi, storage, stream, region, rng)
            except Exception as e:  # noqa: BLE001 -- never costs the headline line
                out["roofline"]["level9_supporting"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(level, list(w))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
