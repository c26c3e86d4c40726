"""P2 Gauss-Seidel / SOR on meshes of several macro-cells: the reference's iterates.

P2ConstantOperator::smooth_sor (src/constant_stencil_operator/P2ConstantOperator.cpp:1267-1330) sweeps macro-vertices, macro-edges
(vertex DoFs along the edge, then its edge DoFs), macro-faces (vertex DoFs, then the edge DoFs with X, XY, Y at every index) and
macro-cells, each primitive with current values on its closure and ghost-layer values elsewhere.  The host layer's smooth_sor
(cell-centric: closure-split operator tables, the P1 shell kernels for the vertex-vertex couplings, hyteg_hip_p2_sor_face_edgedofs_cell
for the edge DoFs inside the macro-faces) is compared with oracle/p2_sor_oracle.GlobalSweep, the same schedule on the global
matrix, which knows nothing about cells' copies: <= 1e-12 on every DoF of every cell, forward and backwards, relax 1 and 0.8,
for the elementwise and the constant-stencil operator."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _setup(mesh, level, constant):
    import torch

    import hostutil as hu
    from hyteg_amd import host
    from oracle import p2_sor_oracle as ps

    assert torch.cuda.is_available()
    v, c = hu.read_msh(hu.MESHES / f"{mesh}.msh")
    G = ps.GlobalSweep(v, c, level)
    st = host.Storage.from_gmsh(hu.MESHES / f"{mesh}.msh")
    gids = [st.local_cell(k)[0] for k in range(st.n_local_cells)]
    A = (host.P2ConstantLaplaceOperator if constant else host.P2ElementwiseLaplaceOperator)(st, level, level)
    A.compute_inverse_diagonal()
    return host, G, st, gids, A


def _upload(G, f, vec, gids, level):
    cells = G.to_cells(vec)
    for k, gid in enumerate(gids):
        f.upload(level, cells[gid][0], cells[gid][1], k)


def _compare(G, f, vec, gids, level, tol=1e-12):
    cells = G.to_cells(vec)
    worst = 0.0
    for k, gid in enumerate(gids):
        gv, ge = f.download(level, k)
        worst = max(worst, np.abs(gv - cells[gid][0]).max(), np.abs(ge - cells[gid][1]).max())
    assert worst < tol, worst
    return worst


@pytest.mark.parametrize("mesh,level,constant", [("regular_octahedron_8el", 2, False), ("regular_octahedron_8el", 3, False),
                                                 ("regular_octahedron_8el", 2, True), ("cube_6el", 2, False), ("cube_6el", 3, True)])
def test_p2_smooth_sor_gives_the_reference_iterates(mesh, level, constant):
    host, G, st, gids, A = _setup(mesh, level, constant)
    rng = np.random.default_rng(17)
    u0, b = rng.standard_normal(G.ndof), rng.standard_normal(G.ndof)
    u, r = host.P2Function(st, "u", level, level), host.P2Function(st, "b", level, level)
    _upload(G, r, b, gids, level)
    for backwards in (False, True):
        for relax in (1.0, 0.8):
            _upload(G, u, u0, gids, level)
            A.smooth_sor(u, r, relax, level, host.Inner, backwards)
            ref = G.sweep(u0, b, relax, backwards)
            _compare(G, u, ref, gids, level)
            # and a second sweep on top (the copies of shared DoFs must have stayed consistent)
            A.smooth_sor(u, r, relax, level, host.Inner, backwards)
            _compare(G, u, G.sweep(ref, b, relax, backwards), gids, level)
    for o in (u, r, A):
        o.close()


def test_p2_gauss_seidel_converges_to_the_discrete_solution():
    """smooth_gs forward / backward alternating: the residual of the inner equations falls monotonically and the exact discrete
    solution (a quadratic: the P2 interpolant IS the finite-element solution) is a fixed point"""
    mesh, level = "regular_octahedron_8el", 2
    host, G, st, gids, A = _setup(mesh, level, False)
    n = 1 << level
    X = np.zeros((G.ndof, 3))
    import hostutil as hu

    v, _ = hu.read_msh(hu.MESHES / f"{mesh}.msh")
    for key, i in G.index.items():
        X[i] = sum(np.asarray(v[a]) * w for a, w in key) / (2 * n)
    q = X[:, 0] ** 2 - X[:, 1] ** 2 + 0.3 * X[:, 0] * X[:, 2]  # harmonic
    u, r = host.P2Function(st, "u", level, level), host.P2Function(st, "b", level, level)
    r.interpolate(0.0, level)
    _upload(G, u, q, gids, level)
    A.smooth_gs(u, r, level, host.Inner)
    _compare(G, u, q, gids, level, 1e-13)
    inner = ~np.array(G.boundary)
    start = q.copy()
    start[inner] = 0.0
    _upload(G, u, start, gids, level)
    for k in range(30):
        A.smooth_sor(u, r, 1.0, level, host.Inner, bool(k & 1))
    cells = [u.download(level, k) for k in range(len(gids))]
    got = G.to_global([cells[gids.index(g)] for g in range(len(gids))])
    assert np.abs(got - q).max() < 0.2 * np.abs(start - q).max()
    for o in (u, r, A):
        o.close()
