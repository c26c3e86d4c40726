"""BASELINE config 5's operator on several ranks: the P1-P1 Stokes operator and its Uzawa smoother (Gauss-Seidel on the velocity
block, shared launches between the components) with the macro-cells of regular_octahedron_8el partitioned over two ranks that
share this box's one GPU must reproduce the single-rank numbers -- over the gloo hook transport and over the peer-to-peer
transport (comm_p2p.hip)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
MESH = ROOT / "hyteg_amd" / "data" / "meshes" / "regular_octahedron_8el.msh"
FUNCS = [lambda x, y, z: np.sin(5 * x) * y, lambda x, y, z: z * np.cos(3 * y), lambda x, y, z: x * y * z, lambda x, y, z: np.sin(2 * x + y - z)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(host, st, level):
    sys.path.insert(0, str(ROOT / "tests"))
    from hostutil import cell_points

    L = host.P1P1StokesOperator(st, level, level)
    x, b, r = (host.P1StokesFunction(st, n, level, level) for n in ("x", "b", "r"))
    flag = host.Inner | host.NeumannBoundary
    for k, f in enumerate(FUNCS):
        for c in range(st.n_local_cells):
            gid, co, nnc = st.local_cell(c)
            P = cell_points(co, level)
            x.components[k].upload_cell(c, level, np.ascontiguousarray(f(P[:, 0], P[:, 1], P[:, 2])))
        x.components[k].interpolate(0.0, level, host.DirichletBoundary)
        b.components[k].interpolate(0.0, level, host.All)

    def residual():
        L.apply(x, r, level, flag)
        r.assign([1.0, -1.0], [b, r], level, flag)
        return np.sqrt(r.dot(r, level, flag))

    uz = host.StokesSolver.uzawa(st, level, level, 0.3, velocity_iterations=2, velocity_smoother=host.GAUSS_SEIDEL)
    res = [residual()]
    for _ in range(3):
        uz.solve(L, x, b, level)
        res.append(residual())
    out = {st.local_cell(c)[0]: [x.components[k].download_cell(c, level) for k in range(4)] for c in range(st.n_local_cells)}
    for o in (uz, x, b, r, L):
        o.close()
    return res, out


def _worker(rank, world, port, level, q, transport):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from hyteg_amd import host
    from hyteg_amd.distributed import DistributedContext

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = host.Storage.from_gmsh(MESH, rank, world)
        st.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx = DistributedContext(st, [level], torch.device("cuda", 0), transport=transport)
        assert ctx.transport == ("p2p" if transport == "p2p" else "hooks"), ctx.transport_note
        out = _run(host, st, level)
        st.check_transport()
        q.put((rank,) + out)
        dist.barrier()
    except BaseException as e:
        q.put(("error", rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["auto", "p2p"])
def test_stokes_operator_and_uzawa_smoother_on_two_ranks(transport):
    import torch
    import torch.multiprocessing as mp

    sys.path.insert(0, str(ROOT))
    from hyteg_amd import host

    assert torch.cuda.is_available()
    level, world = 3, 2
    st = host.Storage.from_gmsh(MESH)
    st.set_stream(torch.cuda.current_stream().cuda_stream)
    ref_res, ref_out = _run(host, st, level)
    assert ref_res[-1] < ref_res[0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, level, q, transport)) for r in range(world)]
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
    for rank, res, out in results:
        assert np.allclose(res, ref_res, rtol=1e-10, atol=0.0), (res, ref_res)
        for gid, comps in out.items():
            for k in range(4):
                scale = max(1.0, np.abs(ref_out[gid][k]).max())
                assert np.abs(comps[k] - ref_out[gid][k]).max() <= 1e-11 * scale, f"component {k} differs on rank {rank}, cell {gid}"
            cells += 1
    assert cells == 8
