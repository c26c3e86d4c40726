"""world_size-2 (and 3) CPU tests of the N>1 path over the gloo backend: every rank builds its own storage and
exchange plan from the global mesh, packs the partial values of shared DoFs exactly as the pack kernel would
(numpy gather on the exported plan), exchanges them with DistributedContext.exchange (all_to_all_single), reduces
in plan order, and must reproduce the single-process sums bit for bit."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
MESH = ROOT / "hyteg_amd" / "data" / "meshes" / "regular_octahedron_8el.msh"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _partials(storage, level):
    """per local cell: a 'partial result' that depends on the physical point AND on the cell"""
    sys.path.insert(0, str(ROOT / "tests"))
    from hostutil import cell_points

    out = []
    for i in range(storage.n_local_cells):
        gid, co, nnc = storage.local_cell(i)
        P = cell_points(co, level)
        out.append(np.ascontiguousarray((1.0 + gid) * (np.sin(3 * P[:, 0]) + P[:, 1] * 7 - P[:, 2] ** 2)))
    return out


def _worker(rank, world, port, level, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from hyteg_amd import host
    from hyteg_amd.distributed import DistributedContext

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = host.Storage.from_gmsh(MESH, rank, world)
        ctx = DistributedContext(st, [level], "cpu")
        arrays = _partials(st, level)
        nloc = st.n_local_cells
        for cls in (0, 1):
            p = ctx.plans[(level, cls)]
            if len(p["peers"]):
                send = ctx.send_tensor(level, cls)
                packed = np.array([arrays[b][o] for b, o in zip(p["send_buf"], p["send_off"])])
                send[: len(packed)] = torch.from_numpy(packed)
                ctx.exchange(level, cls)
                recv = ctx.recv_tensor(level, cls).numpy()
                segs, off = [], 0
                for k in range(len(p["peers"])):
                    segs.append(recv[off: off + int(p["recv_count"][k])])
                    off += int(p["recv_count"][k])
            else:
                segs = []
            bases = arrays + segs
            gp, eb, eo = p["group_ptr"], p["entry_buf"], p["entry_off"]
            for g in range(p["ngroups"]):
                s = 0.0
                for e in range(gp[g], gp[g + 1]):
                    s = bases[eb[e]][eo[e]] if e == gp[g] else s + bases[eb[e]][eo[e]]
                for e in range(gp[g], gp[g + 1]):
                    if eb[e] < nloc:
                        bases[eb[e]][eo[e]] = s
        # dotGlobal hook
        v = (C_double := __import__("ctypes").c_double * 2)(1.0 + rank, 10.0)
        ctx.allreduce_sum(v, 2)
        q.put((rank, [st.local_cell(i)[0] for i in range(nloc)], arrays, list(v)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_over_gloo_reproduces_the_single_process_sums(world):
    sys.path.insert(0, str(ROOT))
    from hyteg_amd import host

    level = 2
    # single-process reference
    st = host.Storage.from_gmsh(MESH)
    ref = _partials(st, level)
    for cls in (0, 1):
        p = st.plan(level, cls)
        gp, eb, eo = p["group_ptr"], p["entry_buf"], p["entry_off"]
        for g in range(p["ngroups"]):
            s = 0.0
            for e in range(gp[g], gp[g + 1]):
                s = ref[eb[e]][eo[e]] if e == gp[g] else s + ref[eb[e]][eo[e]]
            for e in range(gp[g], gp[g + 1]):
                ref[eb[e]][eo[e]] = s
    st.close()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, level, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    seen = 0
    for rank, gids, arrays, red in results:
        assert red == [sum(1.0 + r for r in range(world)), 10.0 * world]
        for gid, arr in zip(gids, arrays):
            assert np.array_equal(arr, ref[gid]), f"rank {rank} cell {gid}"  # bit for bit: fixed summation order
            seen += 1
    assert seen == 8


def test_an_exception_in_a_hook_fails_the_operation_that_called_it():
    """a Python exception inside a ctypes callback used to be printed and dropped, and the host layer went
    on to reduce stale receive buffers.  Hooks return a status now: the C++ layer throws, and the binding re-raises with
    the original exception as its cause."""
    sys.path.insert(0, str(ROOT))
    from hyteg_amd import host

    st = host.Storage.from_gmsh(MESH, 0, 2)  # rank 0 of 2: distributed, so global sums go through the transport
    with pytest.raises(host.HytegHostError, match="no transport"):
        st.allreduce_sum([1.0])

    calls = []

    def bad_allreduce(values, n):
        calls.append(n)
        raise RuntimeError("link down")

    st.set_hooks(lambda level, key: None, lambda level, key: None, bad_allreduce)
    assert st.transport == "hooks"
    with pytest.raises(host.HytegHostError, match="all-reduce hook failed") as ei:
        st.allreduce_sum([1.0, 2.0])
    assert calls == [2]
    assert isinstance(ei.value.__cause__, RuntimeError) and "link down" in str(ei.value.__cause__)

    # a hook that works: values are summed in place (here: doubled, standing in for a second rank)
    def ok_allreduce(values, n):
        for k in range(n):
            values[k] *= 2.0

    st.set_hooks(lambda level, key: None, lambda level, key: None, ok_allreduce)
    assert list(st.allreduce_sum([1.0, 2.5])) == [2.0, 5.0]
    st.close()
