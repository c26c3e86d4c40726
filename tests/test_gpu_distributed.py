"""Two ranks sharing ONE GPU (gloo transport, staged through host memory) run the complete multi-rank host-layer
path with the real kernels -- boundary shares, pack kernel, exchange hooks, reduction kernel, dotGlobal all-reduce,
a full V-cycle -- and must reproduce the single-rank results.  (RCCL itself needs one GPU per rank and is exercised
by the driver's multi-GPU bench; the code path above the transport is identical.)"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
MESH = ROOT / "hyteg_amd" / "data" / "meshes" / "regular_octahedron_8el.msh"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fields(host, storage, level):
    sys.path.insert(0, str(ROOT / "tests"))
    from hostutil import cell_points

    out = []
    for i in range(storage.n_local_cells):
        gid, co, nnc = storage.local_cell(i)
        P = cell_points(co, level)
        out.append((gid, np.ascontiguousarray(np.sin(5 * P[:, 0] + 2 * P[:, 1]) + P[:, 2] * P[:, 0])))
    return out


def _run(host, storage, level, ctx=None):
    """apply + dot + one V(2,2) Jacobi cycle + a forward and a backward Gauss-Seidel sweep; returns {global cell id: arrays}"""
    A = host.P1ConstantOperator(storage, 2, level)
    A.compute_inverse_diagonal()
    u, r, b = (host.P1Function(storage, n, 2, level) for n in ("u", "r", "b"))
    for c, (gid, arr) in enumerate(_fields(host, storage, level)):
        u.upload_cell(c, level, arr)
    u.sync_shared(level, host.All)
    u.interpolate(0.0, level, host.DirichletBoundary)
    A.apply(u, r, level, host.Inner)
    dot = r.dot(r, level, host.Inner)
    applied = {storage.local_cell(c)[0]: r.download_cell(c, level) for c in range(storage.n_local_cells)}
    gmg = host.Solver.gmg(storage, 2, level, smoother=host.JACOBI, relax=2.0 / 3.0, pre=2, post=2)
    gmg.solve(A, u, b, level)
    cycled = {storage.local_cell(c)[0]: u.download_cell(c, level) for c in range(storage.n_local_cells)}
    b.interpolate(1.0, level, host.All)
    A.smooth_sor(u, b, 1.0, level, host.Inner, False)
    A.smooth_sor(u, b, 1.0, level, host.Inner, True)
    swept = {storage.local_cell(c)[0]: u.download_cell(c, level) for c in range(storage.n_local_cells)}
    return applied, dot, cycled, swept


def _worker(rank, world, port, level, q, transport="auto", mesh=MESH):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # one macro-cell per rank + p2p: also at these small levels the share kernel delivers the shares itself (production: only
    # exchanges of >= 60000 values)
    os.environ["HYTEG_AMD_SHARE_SEND_MIN"] = "1"
    import torch
    import torch.distributed as dist

    from hyteg_amd import host
    from hyteg_amd.distributed import DistributedContext

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = host.Storage.from_gmsh(mesh, rank, world)
        st.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx = DistributedContext(st, [2, 3, level] if level > 3 else [2, 3], torch.device("cuda", 0), transport=transport)
        if transport == "p2p":
            # no silent fallback in a test: the peer-to-peer set-up (IPC handles between the two processes) must have worked
            assert ctx.transport == "p2p" and st.transport == "p2p", ctx.transport_note
        out = _run(host, st, level, ctx)
        st.check_transport()  # no device-side wait has timed out
        q.put((rank,) + out)
        dist.barrier()
    except BaseException as e:  # the parent fails at once instead of waiting for the queue
        q.put(("error", rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,transport,mesh", [(2, "auto", "regular_octahedron_8el"), (2, "p2p", "regular_octahedron_8el"),
                                                  (4, "p2p", "regular_octahedron_8el"), (2, "p2p", "pyramid_2el"),
                                                  (4, "p2p", "pyramid_4el")])
def test_ranks_on_one_gpu_reproduce_the_single_rank_results(world, transport, mesh):
    """transport "auto": gloo hooks staged through host memory.  "p2p": the pack kernels store into the other processes'
    IPC-mapped arenas and wait kernels poll sequence numbers (comm_p2p.hip) -- no host synchronisation between the ranks'
    kernels, so this also exercises the slot / sequence protocol under real asynchrony (dots still go through gloo).
    The pyramid meshes give ONE macro-cell per rank (the configuration of bench.py --gpus N)."""
    import torch
    import torch.multiprocessing as mp

    sys.path.insert(0, str(ROOT))
    from hyteg_amd import host

    assert torch.cuda.is_available()
    level = 3
    mesh = ROOT / "hyteg_amd" / "data" / "meshes" / f"{mesh}.msh"
    st = host.Storage.from_gmsh(mesh)
    st.set_stream(torch.cuda.current_stream().cuda_stream)
    if "pyramid" in mesh.name:
        # one macro-cell per rank: the ranks launch per cell, so the single-rank reference must as well (its batched kernels for
        # several small cells sum the boundary shares in another order: equal to rounding, not bit for bit)
        st.set_batch_max_level(-1)
    ref_applied, ref_dot, ref_cycled, ref_swept = _run(host, st, level)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, level, q, transport, mesh), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    for _ in range(world):
        results.append(q.get(timeout=240))
        assert results[-1][0] != "error", results[-1]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cells = 0
    for rank, applied, dot, cycled, swept in results:
        assert abs(dot - ref_dot) <= 1e-12 * abs(ref_dot)  # different partial-sum grouping across ranks
        for gid, arr in applied.items():
            assert np.array_equal(arr, ref_applied[gid]), f"apply differs on rank {rank}, cell {gid}"
            cells += 1
        for gid, arr in cycled.items():
            assert np.allclose(arr, ref_cycled[gid], rtol=1e-11, atol=1e-13), f"V-cycle differs on rank {rank}, cell {gid}"
        for gid, arr in swept.items():
            assert np.allclose(arr, ref_swept[gid], rtol=1e-11, atol=1e-13), f"Gauss-Seidel sweeps differ on rank {rank}, cell {gid}"
    assert cells == st.n_cells


# ---- a rank that shares nothing with anybody must still take part in the (collective) exchange ----
ISOLATED_V = [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1], [3, 0, 0], [4, 0, 0], [3, 1, 0], [3, 0, 1]]
ISOLATED_C = [[0, 1, 2, 3], [1, 2, 3, 4], [5, 6, 7, 8]]  # cells 0 and 1 share a face, cell 2 stands alone


def _run_isolated(host, storage, level):
    A = host.P1ConstantOperator(storage, level, level)
    u, r = host.P1Function(storage, "u", level, level), host.P1Function(storage, "r", level, level)
    for c, (gid, arr) in enumerate(_fields(host, storage, level)):
        u.upload_cell(c, level, arr)
    u.sync_shared(level, host.All)
    A.apply(u, r, level, host.All)
    dot = r.dot(r, level, host.All)
    return {storage.local_cell(c)[0]: r.download_cell(c, level) for c in range(storage.n_local_cells)}, dot


def _worker_isolated(rank, world, port, level, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from hyteg_amd import host
    from hyteg_amd.distributed import DistributedContext

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = host.Storage.from_arrays(ISOLATED_V, ISOLATED_C, rank, world)
        st.set_boundary_type(host.NeumannBoundary)
        st.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx = DistributedContext(st, [level], torch.device("cuda", 0))  # noqa: F841
        q.put((rank,) + _run_isolated(host, st, level))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_a_rank_without_shared_points_takes_part_in_the_exchange():
    import torch
    import torch.multiprocessing as mp

    sys.path.insert(0, str(ROOT))
    from hyteg_amd import host

    level, world = 3, 3
    st = host.Storage.from_arrays(ISOLATED_V, ISOLATED_C)
    st.set_boundary_type(host.NeumannBoundary)
    st.set_stream(torch.cuda.current_stream().cuda_stream)
    ref, ref_dot = _run_isolated(host, st, level)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_isolated, args=(r, world, port, level, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = 0
    for rank, applied, dot in results:
        assert abs(dot - ref_dot) <= 1e-12 * abs(ref_dot)
        for gid, arr in applied.items():
            # one cell per rank runs the per-cell kernels, three local cells the batched ones: other summation order
            assert np.abs(arr - ref[gid]).max() <= 1e-13 * np.abs(ref[gid]).max()
            seen += 1
    assert seen == 3


# ---- the RCCL backend itself (one rank: all this box can hold) --------------------------------------------------
def _worker_rccl(port, level, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from hyteg_amd import host
    from hyteg_amd.distributed import DistributedContext

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        st = host.Storage.from_gmsh(MESH, 0, 1)
        st.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx = DistributedContext(st, [2, 3], dev)  # "nccl" backend on a device: the native RCCL transport
        assert ctx.transport == "rccl" and st.transport == "rccl"
        res = _run(host, st, level, ctx)
        # the C-ABI the C++ transport is made of, on a communicator of its own: a group of ncclSend / ncclRecv on a side
        # stream ordered against the current stream with events (loop-back: one GPU holds one RCCL rank), and the
        # in-place all-reduce of doubles that dotGlobal uses
        from hyteg_amd import capi

        origin = capi.comm_available()
        comm = capi.comm_create(1, 0, capi.comm_unique_id())
        side = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        a = torch.arange(5000, dtype=torch.float64, device=dev) * 0.5
        b = torch.zeros(5000, dtype=torch.float64, device=dev)
        a.mul_(2.0)  # producer on the current stream: the exchange must see its result
        packed, arrived = capi.event_create(), capi.event_create()
        capi.event_record(packed, cur.cuda_stream)
        capi.stream_wait_event(side.cuda_stream, packed)
        capi.comm_exchange(comm, [0, 0], a.data_ptr(), [3000, 1000], b.data_ptr(), [3000, 1000], side.cuda_stream)
        capi.event_record(arrived, side.cuda_stream)
        capi.stream_wait_event(cur.cuda_stream, arrived)
        c = b + 1.0  # consumer on the current stream
        v = torch.tensor([1.5, -2.0], dtype=torch.float64, device=dev)
        capi.comm_allreduce_sum(comm, v.data_ptr(), 2, cur.cuda_stream)
        torch.cuda.synchronize()
        expect = torch.arange(5000, dtype=torch.float64, device=dev)
        native_ok = (bool(torch.equal(c[:4000], expect[:4000] + 1.0)) and float(b[4000:].abs().sum()) == 0.0
                     and v.tolist() == [1.5, -2.0] and "rccl" in origin)
        capi.event_destroy(packed), capi.event_destroy(arrived)
        capi.comm_destroy(comm)
        # the transport call of the exchange hooks with the argument types they use: split all_to_all of f64 device
        # tensors, asynchronous, stream-ordered wait
        send = torch.arange(1000, dtype=torch.float64, device=dev)
        recv = torch.zeros(1000, dtype=torch.float64, device=dev)
        work = dist.all_to_all_single(recv[:700], send[:700], [700], [700], async_op=True)
        work.wait()
        dist.barrier()
        torch.cuda.synchronize()
        ok = bool(torch.equal(recv[:700], send[:700])) and float(recv[700:].abs().sum()) == 0.0
        q.put(res + (ok and native_ok,))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_single_rank():
    """A one-GPU box can hold one RCCL rank, so this checks what can be checked here: librccl resolves (the copy torch
    already holds), the communicator of the C++ transport comes up on the device through DistributedContext, the
    C-ABI's send/recv group + event ordering + all-reduce work on real RCCL (loop-back), torch's process group (used by
    bench.py for the barrier and the timing reduction) works next to it, and the host layer gives the same numbers as
    without any of it."""
    import torch
    import torch.multiprocessing as mp

    sys.path.insert(0, str(ROOT))
    from hyteg_amd import host

    level = 3
    st = host.Storage.from_gmsh(MESH)
    st.set_stream(torch.cuda.current_stream().cuda_stream)
    ref_applied, ref_dot, ref_cycled, ref_swept = _run(host, st, level)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl, args=(_free_port(), level, q), daemon=True)
    p.start()
    applied, dot, cycled, swept, a2a_ok = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert a2a_ok
    assert dot == ref_dot
    for gid in ref_applied:
        assert np.array_equal(applied[gid], ref_applied[gid])
        assert np.array_equal(cycled[gid], ref_cycled[gid])
        assert np.array_equal(swept[gid], ref_swept[gid])


def test_p2p_canary_between_processes_on_one_gpu():
    """hyteg_amd/p2p_canary.py (what bench.py --gpus N runs before it tries the peer-to-peer transport): three processes map
    each other's arenas through HIP IPC, exchange six rounds with the real pack / wait kernels and check every value"""
    sys.path.insert(0, str(ROOT))
    from hyteg_amd import p2p_canary

    tag = f"test_{os.getpid()}_{_free_port()}"
    procs = [p2p_canary.launch(r, 3, 0, tag) for r in range(3)]
    outs = [p2p_canary.finish(p) for p in procs]
    assert all(ok for ok, _ in outs), outs


# ---- a peer that is late: the arrival wait times out and the next host synchronisation point raises -----------------------
def _worker_late_peer(rank, world, port, level, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HYTEG_HIP_P2P_TIMEOUT_MS="300")
    import faulthandler
    import time

    faulthandler.dump_traceback_later(60, exit=True)  # a rank that hangs says where, and ends

    import torch
    import torch.distributed as dist

    from hyteg_amd import host
    from hyteg_amd.distributed import DistributedContext

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = host.Storage.from_gmsh(ROOT / "hyteg_amd" / "data" / "meshes" / "pyramid_2el.msh", rank, world)
        st.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx = DistributedContext(st, [level], torch.device("cuda", 0), transport="p2p")
        assert ctx.transport == "p2p" and st.transport == "p2p", ctx.transport_note
        A = host.P1ConstantOperator(st, level, level)
        u, r = host.P1Function(st, "u", level, level), host.P1Function(st, "r", level, level)
        u.interpolate(1.0, level, host.All)
        # one ordinary apply + dot first: nothing times out, and every kernel has been launched once (a first launch can take
        # longer than the two seconds below)
        A.apply(u, r, level, host.Inner)
        r.dot(r, level, host.Inner)
        dist.barrier()
        # both ranks make the same calls (the plans without peers still go through the collective hooks), but rank 1 starts its
        # apply two seconds late: rank 0's reduce kernel gives up waiting for rank 1's values after 300 ms and lets stale values
        # through -- the dot product (a point where the host waits for the device anyway) must refuse to go on
        if rank == 1:
            time.sleep(2.0)
        A.apply(u, r, level, host.Inner)
        outcome = "no error"
        if rank == 0:
            try:
                r.dot(r, level, host.Inner)
            except host.HytegHostError as e:
                outcome = str(e)
        else:
            r.download_cell(0, level)  # rank 0's values arrived long ago: nothing to report
        q.put((rank, outcome))
        dist.barrier()
    except BaseException as e:  # report before the process group is torn down (which may block while the peer waits)
        import traceback

        q.put((rank, "worker failed: " + repr(e) + " " + traceback.format_exc()[-1500:]))
        raise
    finally:
        dist.destroy_process_group()


def test_p2p_timeout_fails_the_next_dot_product():
    """a timed-out peer-to-peer arrival wait must not be silent -- P2PTransport checks its status word at the host
    synchronisation points that exist anyway (global sums, downloads of cell arrays)"""
    import torch
    import torch.multiprocessing as mp

    assert torch.cuda.is_available()
    level, world = 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_late_peer, args=(r, world, port, level, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = dict(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:  # a deadlocked rank must not keep the test session alive
            if p.is_alive():
                p.kill()
    assert results[1] == "no error"
    assert "timed out" in results[0], results[0]
