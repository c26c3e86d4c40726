"""BASELINE config 4: P2 elementwise Laplace apply() on cube_24el at level 7, storages partitioned over ranks.
 * level 7, all 24 macro-cells on one rank: the DoFs inside a macro-cell against the oracle applied cell by cell (entry by
   entry), the DoFs shared between cells through properties that hold for the exact operator (a harmonic quadratic has zero
   residual at every inner DoF, u^T A u = int |grad u|^2);
 * two ranks (sharing this box's one GPU, gloo transport through the hooks): the vertex- AND edge-DoF shares travel through the
   exchange and every rank reproduces the single-rank result;
 * P2ConstantLaplaceOperator (own stencil assembly, tests/test_gpu_p2_constant.py) agrees with the elementwise operator."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
MESHES = ROOT / "hyteg_amd" / "data" / "meshes"


def _fields(po, hu, st, level, fn):
    out = []
    for c in range(st.n_local_cells):
        gid, co, nnc = st.local_cell(c)
        out.append((gid, fn(hu.cell_points(co, level)), fn(po.edge_midpoints(co, level))))
    return out


def test_p2_apply_cube_24el_level_7():
    import torch

    import hostutil as hu
    from hyteg_amd import host
    from oracle import p1_oracle as po

    assert torch.cuda.is_available()
    level = 7
    st = host.Storage.from_gmsh(MESHES / "cube_24el.msh")
    assert st.n_local_cells == 24
    A = host.P2ElementwiseLaplaceOperator(st, level, level)
    u, r = host.P2Function(st, "u", level, level), host.P2Function(st, "r", level, level)
    # (1) entry-by-entry: smooth non-polynomial field, DoFs inside the macro-cells (their stencil lies in one cell)
    fn = lambda p: np.sin(3.0 * p[:, 0] + p[:, 1]) * np.cos(2.0 * p[:, 2]) + p[:, 0] * p[:, 1]  # noqa: E731
    fields = _fields(po, hu, st, level, fn)
    for c, (gid, fv, fe) in enumerate(fields):
        u.upload(level, fv, fe, c)
    r.interpolate(0.0, level)
    A.apply(u, r, level, host.Inner)
    inner_v = po.slot_of_points(level) == 14
    inner_e = po.edge_classes(level) == 14
    for c in (0, 11, 23):
        gid, co, nnc = st.local_cell(c)
        em = po.p2_cell_element_matrices(np.asarray(co).reshape(12), level)
        ov, oe = po.p2_elementwise_apply_cell(np.zeros(po.cell_size(level)), np.zeros(po.edge_array_size(level)), fields[c][1], fields[c][2],
                                              level, em, 1.0, 0, 0x7FFF)
        gv, ge = r.download(level, c)
        # the results are differences of terms of size |element matrix entry| * |u| (A u = O(h^3) for a smooth u): the
        # error bound is relative to the terms, not to the cancelled result
        scale = np.abs(em).max() * max(np.abs(fields[c][1]).max(), np.abs(fields[c][2]).max())
        assert np.abs(gv[inner_v] - ov[inner_v]).max() <= 1e-12 * scale
        assert np.abs(ge[inner_e] - oe[inner_e]).max() <= 1e-12 * scale
    # (2) a harmonic quadratic is reproduced exactly by P2: zero residual at every inner DoF, shared ones included
    harmonic = lambda p: p[:, 0] ** 2 - 0.5 * p[:, 1] ** 2 - 0.5 * p[:, 2] ** 2 + p[:, 1] * p[:, 2] - p[:, 0]  # noqa: E731
    for c, (gid, fv, fe) in enumerate(_fields(po, hu, st, level, harmonic)):
        u.upload(level, fv, fe, c)
    r.interpolate(0.0, level)
    A.apply(u, r, level, host.Inner)
    assert np.sqrt(r.dot(r, level, host.Inner)) < 1e-9  # a wrong stencil entry would give O(h) per DoF: ~60 over the 6e7 DoFs
    # (3) energy: u^T A u = int_cube |grad x^2|^2 = 4/3, every DoF counted once (no Dirichlet DoFs)
    st.set_boundary_type(host.NeumannBoundary)
    for c, (gid, fv, fe) in enumerate(_fields(po, hu, st, level, lambda p: p[:, 0] ** 2)):
        u.upload(level, fv, fe, c)
    A.apply(u, r, level, host.All)
    assert abs(u.dot(r, level, host.All) - 4.0 / 3.0) < 1e-10
    for o in (u, r, A, st):
        o.close()


def test_p2_constant_operator_gives_the_elementwise_numbers():
    """P2ConstantOperator::apply (four constant-stencil sub-operators, P2ConstantOperator.cpp:100-112) against the oracle's
    literal restatement of the elementwise loops, on a skew tetrahedron and on several cells"""
    import torch

    import hostutil as hu
    from hyteg_amd import host
    from oracle import p1_oracle as po

    assert torch.cuda.is_available()
    level = 3
    st = host.Storage.from_gmsh(MESHES / "regular_octahedron_8el.msh")
    Ac, Ae = host.P2ConstantLaplaceOperator(st, level, level), host.P2ElementwiseLaplaceOperator(st, level, level)
    u, rc, re_ = (host.P2Function(st, n, level, level) for n in ("u", "rc", "re"))
    fn = lambda p: np.sin(4.0 * p[:, 0]) + p[:, 1] * p[:, 2] ** 2  # noqa: E731
    fields = _fields(po, hu, st, level, fn)
    for c, (gid, fv, fe) in enumerate(fields):
        u.upload(level, fv, fe, c)
    for r_ in (rc, re_):
        r_.interpolate(0.0, level)
    Ac.apply(u, rc, level, host.Inner)
    Ae.apply(u, re_, level, host.Inner)
    inner_v, inner_e = po.slot_of_points(level) == 14, po.edge_classes(level) == 14
    for c in range(st.n_local_cells):
        (cv, ce), (ev, ee) = rc.download(level, c), re_.download(level, c)
        assert np.abs(cv - ev).max() < 1e-13 * np.abs(ev).max() and np.abs(ce - ee).max() < 1e-13 * np.abs(ee).max()  # two assemblies of one operator
        gid, co, nnc = st.local_cell(c)
        em = po.p2_cell_element_matrices(np.asarray(co).reshape(12), level)
        ov, oe = po.p2_elementwise_apply_cell(np.zeros(po.cell_size(level)), np.zeros(po.edge_array_size(level)), fields[c][1], fields[c][2],
                                              level, em, 1.0, 0, 0x7FFF)
        scale = np.abs(em).max() * max(np.abs(fields[c][1]).max(), np.abs(fields[c][2]).max())
        assert np.abs(cv[inner_v] - ov[inner_v]).max() <= 1e-12 * scale and np.abs(ce[inner_e] - oe[inner_e]).max() <= 1e-12 * scale
    for o in (u, rc, re_, Ac, Ae, st):
        o.close()


# ---- two ranks ------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_p2(host, po, hu, st, level):
    A = host.P2ElementwiseLaplaceOperator(st, level, level)
    u, r = host.P2Function(st, "u", level, level), host.P2Function(st, "r", level, level)
    fn = lambda p: np.sin(3.0 * p[:, 0] + p[:, 1]) * np.cos(2.0 * p[:, 2]) + p[:, 0] * p[:, 1]  # noqa: E731
    gids = []
    for c, (gid, fv, fe) in enumerate(_fields(po, hu, st, level, fn)):
        u.upload(level, fv, fe, c)
        gids.append(gid)
    r.interpolate(0.0, level)
    A.apply(u, r, level, host.Inner)
    out = {gid: r.download(level, c) for c, gid in enumerate(gids)}
    dot = r.dot(r, level, host.Inner)
    # Add mode goes through the temporary + exchange path
    A.apply(u, r, level, host.Inner, host.Add)
    out2 = {gid: r.download(level, c) for c, gid in enumerate(gids)}
    # a Gauss-Seidel sweep forward and one backward (vertex DoFs by the P1 machinery, shared edge DoFs by owner-coloured sweeps with
    # summed shares and synchronised copies, cell edge DoFs by type): the partition over ranks must not change a bit of it beyond
    # the summation order of the shares; the result rides along in out2's place holder for u
    A.compute_inverse_diagonal()
    r.interpolate(0.0, level)
    A.smooth_sor(u, r, 1.0, level, host.Inner, False)
    A.smooth_sor(u, r, 1.0, level, host.Inner, True)
    out3 = {gid: u.download(level, c) for c, gid in enumerate(gids)}
    for o in (u, r, A):
        o.close()
    return out, dot, (out2, out3)


def _worker(rank, world, port, level, q, transport="auto"):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    import hostutil as hu
    from hyteg_amd import host
    from hyteg_amd.distributed import DistributedContext
    from oracle import p1_oracle as po

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = host.Storage.from_gmsh(MESHES / "cube_6el.msh", rank, world)
        st.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx = DistributedContext(st, [level], torch.device("cuda", 0), dof_kinds=(0, 1), transport=transport)  # vertex- and edge-DoF plans
        assert any(k[1] >= 2 for k in ctx.buffers), "edge-DoF plans must have peers on this mesh"
        assert ctx.transport == ("p2p" if transport == "p2p" else "hooks"), ctx.transport_note
        out = _run_p2(host, po, hu, st, level)
        st.check_transport()
        q.put((rank,) + out)
        dist.barrier()
    except BaseException as e:
        q.put(("error", rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["auto", "p2p"])
def test_p2_apply_on_two_ranks_reproduces_the_single_rank_result(transport):
    """transport "p2p": vertex- AND edge-DoF shares are stored straight into the other process's IPC-mapped arena (comm_p2p.hip)"""
    import torch
    import torch.multiprocessing as mp

    import hostutil as hu
    from hyteg_amd import host
    from oracle import p1_oracle as po

    assert torch.cuda.is_available()
    level = 3
    st = host.Storage.from_gmsh(MESHES / "cube_6el.msh")
    ref, ref_dot, (ref2, ref3) = _run_p2(host, po, hu, st, level)
    st.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, level, q, transport), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    results = []
    for _ in range(2):
        results.append(q.get(timeout=120))
        assert results[-1][0] != "error", results[-1]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = 0
    for rank, out, dot, (out2, out3) in results:
        assert abs(dot - ref_dot) <= 1e-12 * ref_dot  # dotGlobal: all-reduce over the ranks
        for gid in out:
            for got, want in ((out[gid], ref[gid]), (out2[gid], ref2[gid]), (out3[gid], ref3[gid])):
                scale = max(np.abs(want[0]).max(), np.abs(want[1]).max())
                assert np.abs(got[0] - want[0]).max() <= 1e-13 * scale
                assert np.abs(got[1] - want[1]).max() <= 1e-13 * scale
            seen += 1
    assert seen == 6
