"""P1-P1 Stokes composition (BASELINE config 5's operator): the blocks of
src/mixed_operator/P1P1StokesOperator.hpp:51-64 as P1ConstantOperator< Form > instances, the composite apply, the Uzawa
smoother (src/hyteg/solvers/UzawaSmoother.hpp:262-288) and the reference's own known answer
tests/hyteg/convergence/P1P1Stokes3DUzawaConvergenceTest.cpp.

The oracle side: oracle/p1_oracle.c restates the element matrices of div / divT / PSPG; tests/test_oracle_pins.py pins them
(and the stencils assembled from them) to the reference's generated FEniCS code compiled in place (oracle/_ref).  Here the
GPU results are compared with that oracle applied cell by cell (tests/hostutil.py MultiCellOracle)."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
MESHES = ROOT / "hyteg_amd" / "data" / "meshes"


@pytest.fixture(scope="module")
def env():
    import torch

    from hyteg_amd import capi, host
    from oracle import p1_oracle as po

    assert torch.cuda.is_available()
    capi.lib()
    host.lib()
    return torch, capi, host, po


def _rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


FORMS = list(range(2, 9))  # div x/y/z, divT x/y/z, PSPG


@pytest.mark.parametrize("mesh", ["regular_octahedron_8el", "pyramid_tilted_4el"])
@pytest.mark.parametrize("form", FORMS)
def test_block_stencils_match_the_oracle(env, mesh, form):
    """inner stencil and the 14 partial stencils of every cell against the oracle's assembly (pinned to the reference's
    FEniCS element matrices in tests/test_oracle_pins.py)"""
    torch, capi, host, po = env
    st = host.Storage.from_gmsh(MESHES / f"{mesh}.msh")
    for level in (2, 4):
        A = host.P1ConstantOperator(st, level, level, form)
        for i in range(st.n_local_cells):
            gid, co, nnc = st.local_cell(i)
            inner, slots = A.stencils(gid, level)
            w = po.assemble_cell_stencil(co, level, form)
            ws = po.assemble_cell_slot_stencils(co, level, form)
            scale = np.abs(w).max()
            assert scale > 0
            assert np.abs(np.asarray(inner) - w).max() <= 1e-14 * scale
            assert np.abs(np.asarray(slots).reshape(14, 15) - np.asarray(ws).reshape(14, 15)).max() <= 1e-14 * scale
        A.close()
    st.close()


@pytest.mark.parametrize("form", FORMS)
def test_div_divt_pspg_annihilate_constants(env, form):
    """row sums: every block has zero row sum at inner points (the gradients of the four basis functions of a micro-cell
    sum to zero), so applying it to a constant gives zero -- for divT only the sum over the three directions of a closed
    star vanishes, which is the same statement for the assembled stencil"""
    torch, capi, host, po = env
    from hostutil import download

    st = host.Storage.from_gmsh(MESHES / "regular_octahedron_8el.msh")
    level = 3
    A = host.P1ConstantOperator(st, level, level, form)
    src, dst = host.P1Function(st, "s", level, level), host.P1Function(st, "d", level, level)
    src.interpolate(3.5, level, host.All)
    dst.interpolate(7.0, level, host.All)
    A.apply(src, dst, level, host.Inner)
    gid, co, nnc = st.local_cell(0)
    scale = np.abs(po.assemble_cell_stencil(co, level, form)).max() * 3.5
    for c, arr in enumerate(download(dst, level)):
        sel = ((st.mask(c, host.Inner) >> po.slot_of_points(level)) & 1).astype(bool)
        assert np.abs(arr[sel]).max() <= 1e-13 * scale
        assert np.all(arr[~sel] == 7.0)
    for o in (src, dst, A, st):
        o.close()


@pytest.mark.parametrize("mesh", ["regular_octahedron_8el", "cube_6el"])
@pytest.mark.parametrize("batch", [6, -1])
def test_stokes_apply_is_the_composition_of_the_oracle_blocks(env, mesh, batch):
    """P1P1StokesOperator::apply: dst.uvw = lapl src.uvw + divT src.p; dst.p = div src.uvw + pspg src.p, velocity on the
    points `flag` selects under the storage's (Dirichlet) boundary types, pressure on ALL points (createAllInnerBC)"""
    torch, capi, host, po = env
    from hostutil import MultiCellOracle, download, upload

    level = 3
    st = host.Storage.from_gmsh(MESHES / f"{mesh}.msh")
    st.set_batch_max_level(batch)
    mo = MultiCellOracle(st)
    L = host.P1P1StokesOperator(st, level, level)
    src, dst = host.P1StokesFunction(st, "src", level, level), host.P1StokesFunction(st, "dst", level, level)
    fields = [lambda x, y, z: np.sin(3 * x + y) + z * z, lambda x, y, z: x * y - np.cos(2 * z), lambda x, y, z: x + 2 * y * z,
              lambda x, y, z: np.sin(4 * x * y) + z]
    src_h = [mo.interpolate(f, level) for f in fields]
    rng = np.random.default_rng(5)
    dst0 = []
    for k in range(4):
        d = [rng.random(po.cell_size(level)) for _ in src_h[0]]
        mo.sync(d, level, host.All)
        dst0.append(d)
    for k in range(4):
        upload(src.components[k], src_h[k], level)
        upload(dst.components[k], dst0[k], level)
    flag = host.Inner | host.NeumannBoundary
    L.apply(src, dst, level, flag)
    zeros = lambda: [np.zeros(po.cell_size(level)) for _ in src_h[0]]  # noqa: E731
    for k in range(3):  # velocity rows
        ref = mo.apply(src_h[k], [d.copy() for d in dst0[k]], level, flag, po.FORM_LAPLACE)
        add = mo.apply(src_h[3], zeros(), level, flag, po.FORM_DIVT_X + k)
        got = download(dst.components[k], level)
        for c, (g, r_, a, d0) in enumerate(zip(got, ref, add, dst0[k])):
            sel = ((st.mask(c, flag) >> po.slot_of_points(level)) & 1).astype(bool)
            assert np.array_equal(g[~sel], d0[~sel])
            assert _rel(g[sel], (r_ + a)[sel]) < 1e-12
    ref = mo.apply(src_h[3], zeros(), level, host.All, po.FORM_PSPG)
    for k in range(3):
        part = mo.apply(src_h[k], zeros(), level, host.All, po.FORM_DIV_X + k)
        ref = [r_ + p_ for r_, p_ in zip(ref, part)]
    got = download(dst.p, level)
    for g, r_ in zip(got, ref):
        assert _rel(g, r_) < 1e-12  # every pressure point is written
    for o in (src, dst, L, st):
        o.close()


def test_project_mean(env):
    torch, capi, host, po = env
    from hostutil import MultiCellOracle, download, upload

    level = 3
    st = host.Storage.from_gmsh(MESHES / "regular_octahedron_8el.msh")
    mo = MultiCellOracle(st)
    p = host.P1Function(st, "p", level, level)
    p.set_all_inner()
    h = mo.interpolate(lambda x, y, z: 3.0 + x - 2 * y * z, level)
    upload(p, h, level)
    ones = [np.ones_like(a) for a in h]
    count = mo.dot(ones, ones, level, host.All)
    mean = mo.dot(h, ones, level, host.All) / count
    host.project_mean(p, level)
    for g, a in zip(download(p, level), h):
        assert np.abs(g - (a - mean)).max() < 1e-14
    one = host.P1Function(st, "one", level, level)
    one.interpolate(1.0, level, host.All)
    assert abs(p.dot(one, level, host.All)) < 1e-12 * count
    for o in (p, one, st):
        o.close()


def test_uzawa_smoother_reduces_the_stokes_residual(env):
    """one application of the smoother is what UzawaSmoother.hpp:262-288 prescribes: checked against the same sequence
    written with the scalar operators, and repeated applications reduce the residual of a homogeneous problem"""
    torch, capi, host, po = env
    from hostutil import MultiCellOracle, upload

    level = 3
    st = host.Storage.from_gmsh(MESHES / "regular_octahedron_8el.msh")
    mo = MultiCellOracle(st)
    L = host.P1P1StokesOperator(st, level, level)
    x, b, r = (host.P1StokesFunction(st, n, level, level) for n in ("x", "b", "r"))
    flag = host.Inner | host.NeumannBoundary
    for k, f in enumerate([lambda x_, y, z: np.sin(5 * x_) * y, lambda x_, y, z: z * np.cos(3 * y), lambda x_, y, z: x_ * y * z,
                           lambda x_, y, z: np.sin(2 * x_ + y - z)]):
        upload(x.components[k], mo.interpolate(f, level), level)
        x.components[k].interpolate(0.0, level, host.DirichletBoundary)
        b.components[k].interpolate(0.0, level, host.All)
    host.project_mean(x.p, level)

    def residual_norm():
        L.apply(x, r, level, flag)
        r.assign([1.0, -1.0], [b, r], level, flag)
        return np.sqrt(r.dot(r, level, flag))

    uz = host.StokesSolver.uzawa(st, level, level, 0.3, velocity_iterations=2, velocity_smoother=host.GAUSS_SEIDEL)
    res = [residual_norm()]
    for _ in range(6):
        uz.solve(L, x, b, level)
        res.append(residual_norm())
    assert all(res[i + 1] < res[i] for i in range(6)), res
    assert res[-1] < 0.5 * res[0], res
    for o in (uz, x, b, r, L, st):
        o.close()


def test_reference_known_answer_p1p1_stokes_3d_uzawa_convergence(env):
    """tests/hyteg/convergence/P1P1Stokes3DUzawaConvergenceTest.cpp: cube_24el, levels 2..5, V(3,3) with smoothing increment
    2, Uzawa( 0.3 ) with Gauss-Seidel on the velocity block, exact coarse-grid solve, colliding-flow boundary data:
    residual reduction < 0.14 per cycle in each of 3 cycles (:178), final discrete L2 errors u+v+w < 2.8e-3, p < 0.13,
    residual < 4e-6 (:197-199)."""
    torch, capi, host, po = env
    from hostutil import MultiCellOracle, download, upload

    min_level, max_level = 2, 5
    st = host.Storage.from_gmsh(MESHES / "cube_24el.msh")
    mo = MultiCellOracle(st)
    L = host.P1P1StokesOperator(st, min_level, max_level)
    u, f, r, exact, err = (host.P1StokesFunction(st, n, min_level, max_level) for n in ("u", "f", "r", "uExact", "err"))
    cf = [lambda x, y, z: 20.0 * x * y ** 3, lambda x, y, z: 5.0 * x ** 4 - 5.0 * y ** 4, lambda x, y, z: 0.0 * x,
          lambda x, y, z: 60.0 * x ** 2 * y - 20.0 * y ** 3]
    for fn in (u, f, r, exact, err):
        for k in range(4):
            fn.components[k].interpolate(0.0, max_level, host.All)
    for k in range(4):
        upload(exact.components[k], mo.interpolate(cf[k], max_level), max_level)
    for k in range(3):  # u.uvw().interpolate( ..., DirichletBoundary )
        tmp = host.P1Function(st, "tmp", max_level, max_level)
        upload(tmp, mo.interpolate(cf[k], max_level), max_level)
        u.components[k].assign([1.0], [tmp], max_level, host.DirichletBoundary)
        tmp.close()
    one = host.P1Function(st, "one", max_level, max_level)
    one.interpolate(1.0, max_level, host.All)
    ndofs = one.dot(one, max_level, host.All)  # numberOfGlobalDoFs< P1FunctionTag >
    flag = host.Inner | host.NeumannBoundary

    def residual():
        L.apply(u, r, max_level, flag)
        return np.sqrt(r.dot(r, max_level, host.All) / (4.0 * ndofs))

    smoother = host.StokesSolver.uzawa(st, min_level, max_level, 0.3, velocity_iterations=2, velocity_smoother=host.GAUSS_SEIDEL)
    gmg = host.StokesSolver.gmg(st, smoother, min_level, max_level, pre=3, post=3, increment=2, project_mean_after_restriction=True)
    # r holds values from before on the Dirichlet points: the reference's r is freshly zero there
    last = residual()
    rates = []
    for _ in range(3):
        gmg.solve(L, u, f, max_level)
        host.project_mean(u.p, max_level)
        host.project_mean(exact.p, max_level)
        res = residual()
        rates.append(res / last)
        last = res
    err.assign([1.0, -1.0], [u, exact], max_level, host.All)
    e = [np.sqrt(err.components[k].dot(err.components[k], max_level, host.All) / ndofs) for k in range(4)]
    assert all(rate < 1.4e-1 for rate in rates), rates
    assert e[0] + e[1] + e[2] < 2.8e-3, e
    assert e[3] < 0.13, e
    assert last < 4.0e-6, last
    for o in (gmg, smoother, one, u, f, r, exact, err, L, st):
        o.close()
