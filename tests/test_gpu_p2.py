"""GPU parity of the P2 elementwise operator on one macro-cell against the CPU restatement of
P2ElementwiseOperator::gemv (oracle/p1_oracle.c ho_p2_elementwise_apply_cell), through the C-ABI."""
import numpy as np
import pytest

from conftest import OCT_TET, REF_TET, SKEW_TET

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from hyteg_amd import capi
    from oracle import p1_oracle as po

    assert torch.cuda.is_available()
    capi.lib()
    return torch, capi, po


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("tet", [REF_TET, SKEW_TET])
def test_p2_elementwise_apply_matches_the_oracle(env, level, tet):
    torch, capi, po = env
    nv, ne = po.cell_size(level), po.edge_array_size(level)
    assert ne == capi.p2_edge_array_size(level)
    em = po.p2_cell_element_matrices(np.asarray(tet, dtype=np.float64).reshape(12), level)
    rng = np.random.default_rng(level)
    sv, se, dv0, de0 = rng.standard_normal(nv), rng.standard_normal(ne), rng.standard_normal(nv), rng.standard_normal(ne)
    dem = _dev(torch, capi.p2_build_operator_table(em))
    for mask, update, alpha in ((0x7FFF, 0, 1.0), (1 << 14, 0, 1.0), (0x7FFF, 1, -0.5), (0x4000 | 0x2A5, 0, 2.0), (0x3FFF, 1, 1.0)):
        wv, we = po.p2_elementwise_apply_cell(dv0.copy(), de0.copy(), sv, se, level, em, alpha, update, mask)
        dsv, dse, ddv, dde = _dev(torch, sv), _dev(torch, se), _dev(torch, dv0), _dev(torch, de0)
        capi.p2_elementwise_apply_cell(ddv.data_ptr(), dde.data_ptr(), dsv.data_ptr(), dse.data_ptr(), level, dem.data_ptr(), alpha, update, mask)
        torch.cuda.synchronize()
        gv, ge = ddv.cpu().numpy(), dde.cpu().numpy()
        scale = max(np.abs(wv).max(), np.abs(we).max(), 1.0)
        assert np.abs(gv - wv).max() <= 1e-13 * scale and np.abs(ge - we).max() <= 1e-13 * scale, (level, hex(mask), update)
        # unselected DoFs are untouched
        sel_v = ((mask >> po.slot_of_points(level)) & 1).astype(bool)
        sel_e = ((mask >> po.edge_classes(level)) & 1).astype(bool)
        assert np.array_equal(gv[~sel_v], dv0[~sel_v]) and np.array_equal(ge[~sel_e], de0[~sel_e])


def test_p2_laplace_known_answers_level_6(env):
    """energy of x^2 on the reference tetrahedron is int |grad x^2|^2 = 1/15; harmonic quadratics have zero inner residual;
    the operator is symmetric (the properties the reference's P2 convergence tests rest on)"""
    torch, capi, po = env
    import hostutil as hu

    level = 6
    co = np.asarray(REF_TET, dtype=np.float64).reshape(12)
    em = _dev(torch, capi.p2_build_operator_table(po.p2_cell_element_matrices(co, level)))
    pv, pe = hu.cell_points(co, level), po.edge_midpoints(co, level)

    def apply(uv, ue):
        duv, due = _dev(torch, uv), _dev(torch, ue)
        dv, de = torch.zeros_like(duv), torch.zeros_like(due)
        capi.p2_elementwise_apply_cell(dv.data_ptr(), de.data_ptr(), duv.data_ptr(), due.data_ptr(), level, em.data_ptr())
        torch.cuda.synchronize()
        return dv.cpu().numpy(), de.cpu().numpy()

    uv, ue = pv[:, 0] ** 2, pe[:, 0] ** 2
    av, ae = apply(uv, ue)
    assert abs(uv @ av + ue @ ae - 1.0 / 15.0) < 1e-12
    hv, he = pv[:, 0] ** 2 - pv[:, 1] ** 2 + 3 * pv[:, 2], pe[:, 0] ** 2 - pe[:, 1] ** 2 + 3 * pe[:, 2]
    rv, re = apply(hv, he)
    inner_v, inner_e = po.slot_of_points(level) == 14, po.edge_classes(level) == 14
    assert np.abs(rv[inner_v]).max() < 1e-13 and np.abs(re[inner_e]).max() < 1e-13
    rng = np.random.default_rng(1)
    xv, xe, yv, ye = rng.standard_normal(len(pv)), rng.standard_normal(len(pe)), rng.standard_normal(len(pv)), rng.standard_normal(len(pe))
    axv, axe = apply(xv, xe)
    ayv, aye = apply(yv, ye)
    lhs, rhs = yv @ axv + ye @ axe, xv @ ayv + xe @ aye
    assert abs(lhs - rhs) < 1e-11 * abs(lhs)


def test_p2_host_layer_solves_a_dirichlet_problem_exactly(env):
    """P2Function / P2ElementwiseLaplaceOperator / CGSolver through the host layer: a quadratic harmonic function is in the
    P2 space, so with its values as Dirichlet data the discrete solution IS the function (the statement the reference's
    P2ElementwiseCGConvergenceTest makes for the discretisation error, here with error zero up to the solver tolerance)."""
    torch, capi, po = env
    import hostutil as hu
    from hyteg_amd import host

    host.lib()
    level = 3
    st = host.Storage.from_gmsh(hu.MESHES / "tet_1el.msh")
    gid, co, nnc = st.local_cell(0)
    A = host.P2ElementwiseLaplaceOperator(st, level, level)
    em = A.element_matrices(level)
    want = po.p2_cell_element_matrices(np.asarray(co).reshape(12), level)
    assert np.abs(em - want).max() <= 1e-14 * np.abs(want).max()

    fn = lambda p: p[:, 0] ** 2 - 0.5 * p[:, 1] ** 2 - 0.5 * p[:, 2] ** 2 + p[:, 0] * p[:, 2] + 1.0  # noqa: E731  (harmonic)
    ev, ee = fn(hu.cell_points(co, level)), fn(po.edge_midpoints(co, level))
    inner_v, inner_e = po.slot_of_points(level) == 14, po.edge_classes(level) == 14
    x, b, r, exact = (host.P2Function(st, n, level, level) for n in ("x", "b", "r", "exact"))
    exact.upload(level, ev, ee)
    x.upload(level, np.where(inner_v, 0.0, ev), np.where(inner_e, 0.0, ee))  # Dirichlet data, zero initial guess inside
    # apply on the exact solution: zero residual on inner DoFs, nothing written elsewhere
    r.interpolate(3.0, level)
    A.apply(exact, r, level, host.Inner)
    rv, re = r.download(level)
    assert np.abs(rv[inner_v]).max() < 1e-12 and np.abs(re[inner_e]).max() < 1e-12
    assert np.all(rv[~inner_v] == 3.0) and np.all(re[~inner_e] == 3.0)
    its = A.cg_solve(x, b, level, 500, 1e-13)
    xv, xe = x.download(level)
    assert 0 < its < 500
    assert np.abs(xv - ev).max() < 1e-10 and np.abs(xe - ee).max() < 1e-10
    # dot product over both parts, flag-filtered
    d = exact.dot(exact, level, host.Inner)
    assert abs(d - ((ev[inner_v] ** 2).sum() + (ee[inner_e] ** 2).sum())) < 1e-11 * d
    for o in (x, b, r, exact, A, st):
        o.close()


# ---- several macro-cells (one rank): shares of the DoFs on common faces / edges are summed by the additive exchange ----
def _p2_fields(host, st, level, fn, po):
    import hostutil as hu

    out = []
    for c in range(st.n_local_cells):
        gid, co, nnc = st.local_cell(c)
        out.append((fn(hu.cell_points(co, level)), fn(po.edge_midpoints(co, level))))
    return out


@pytest.mark.parametrize("mesh", ["cube_6el", "regular_octahedron_8el", "pyramid_2el"])
def test_p2_multi_cell_apply_matches_the_cell_by_cell_oracle(env, mesh):
    """every cell's contributions from the oracle, shared DoFs matched through global keys (end points for edge DoFs) and summed"""
    torch, capi, po = env
    import hostutil as hu
    from hyteg_amd import host

    level = 2
    v, c = hu.read_msh(hu.MESHES / f"{mesh}.msh")
    glob = hu.GlobalSweepOracle(v, c, level)  # only its global vertex numbering is used
    st = host.Storage.from_gmsh(hu.MESHES / f"{mesh}.msh")
    A = host.P2ElementwiseLaplaceOperator(st, level, level)
    src, dst = host.P2Function(st, "src", level, level), host.P2Function(st, "dst", level, level)
    rng = np.random.default_rng(5)
    ncell, N = st.n_local_cells, (1 << level) + 1
    ec = po.edge_coords(level)
    ends = np.array([[[0, 0, 0], [1, 0, 0]], [[0, 0, 0], [0, 1, 0]], [[0, 0, 0], [0, 0, 1]], [[1, 0, 0], [0, 1, 0]],
                     [[1, 0, 0], [0, 0, 1]], [[0, 1, 0], [0, 0, 1]], [[0, 1, 0], [1, 0, 1]]])
    # global values: one random number per physical DoF
    gv = rng.standard_normal(glob.ndof)
    ekeys = []
    for cell in range(ncell):
        keys = []
        for x, y, z, o in ec:
            a = glob.gidx[cell][po.cell_index(level, *(np.array([x, y, z]) + ends[o][0]))]
            b = glob.gidx[cell][po.cell_index(level, *(np.array([x, y, z]) + ends[o][1]))]
            keys.append((min(a, b), max(a, b)))
        ekeys.append(keys)
    evals = {}
    for keys in ekeys:
        for k in keys:
            evals.setdefault(k, rng.standard_normal())
    for cell in range(ncell):
        src.upload(level, gv[glob.gidx[cell]], np.array([evals[k] for k in ekeys[cell]]), cell)
    flag = host.Inner
    dst.interpolate(5.0, level)
    A.apply(src, dst, level, flag)
    # oracle: per-cell contributions, summed over copies
    tot_v, tot_e = np.zeros(glob.ndof), {}
    parts = []
    for cell in range(ncell):
        gid, co, nnc = st.local_cell(cell)
        em = po.p2_cell_element_matrices(np.asarray(co).reshape(12), level)
        mask = st.mask(cell, flag)
        pv, pe = po.p2_elementwise_apply_cell(np.zeros(po.cell_size(level)), np.zeros(po.edge_array_size(level)), gv[glob.gidx[cell]],
                                              np.array([evals[k] for k in ekeys[cell]]), level, em, 1.0, 0, mask)
        parts.append((pv, pe, mask))
        np.add.at(tot_v, glob.gidx[cell], pv)
        for k, val in zip(ekeys[cell], pe):
            tot_e[k] = tot_e.get(k, 0.0) + val
    scale = max(np.abs(tot_v).max(), max(abs(x) for x in tot_e.values()))
    for cell in range(ncell):
        gv_, ge_ = dst.download(level, cell)
        mask = parts[cell][2]
        sel_v = ((mask >> po.slot_of_points(level)) & 1).astype(bool)
        sel_e = ((mask >> po.edge_classes(level)) & 1).astype(bool)
        want_v = tot_v[glob.gidx[cell]]
        want_e = np.array([tot_e[k] for k in ekeys[cell]])
        assert np.abs(gv_[sel_v] - want_v[sel_v]).max() <= 1e-12 * scale
        assert np.abs(ge_[sel_e] - want_e[sel_e]).max() <= 1e-12 * scale
        assert np.all(gv_[~sel_v] == 5.0) and np.all(ge_[~sel_e] == 5.0)
    for o in (src, dst, A, st):
        o.close()


def test_p2_on_the_unit_cube_known_answers_and_cg(env):
    """cube_6el = the unit cube: u^T A u = int |grad u|^2 (4/3 for x^2, every DoF counted once); a harmonic quadratic has zero
    residual at every inner DoF, also on the faces between the macro-cells; CG recovers it from its boundary values"""
    torch, capi, po = env
    import hostutil as hu
    from hyteg_amd import host

    level = 3
    st = host.Storage.from_gmsh(hu.MESHES / "cube_6el.msh")
    A = host.P2ElementwiseLaplaceOperator(st, level, level)
    u, r, x, b = (host.P2Function(st, n, level, level) for n in ("u", "r", "x", "b"))
    # energy with every DoF in the operator (Neumann boundary type: no DoF is excluded)
    st.set_boundary_type(host.NeumannBoundary)
    for cell, (fv, fe) in enumerate(_p2_fields(host, st, level, lambda p: p[:, 0] ** 2, po)):
        u.upload(level, fv, fe, cell)
    A.apply(u, r, level, host.All)
    assert abs(u.dot(r, level, host.All) - 4.0 / 3.0) < 1e-11
    # Dirichlet problem
    st.set_boundary_type(host.DirichletBoundary)
    harmonic = lambda p: p[:, 0] ** 2 - 0.5 * p[:, 1] ** 2 - 0.5 * p[:, 2] ** 2 + p[:, 1] * p[:, 2] - p[:, 0]  # noqa: E731
    exact = _p2_fields(host, st, level, harmonic, po)
    for cell, (fv, fe) in enumerate(exact):
        u.upload(level, fv, fe, cell)
    r.interpolate(0.0, level)
    A.apply(u, r, level, host.Inner)
    assert r.dot(r, level, host.Inner) < 1e-22
    for cell, (fv, fe) in enumerate(exact):
        mask = st.mask(cell, host.Inner)
        iv = ((mask >> po.slot_of_points(level)) & 1).astype(bool)
        ie = ((mask >> po.edge_classes(level)) & 1).astype(bool)
        x.upload(level, np.where(iv, 0.0, fv), np.where(ie, 0.0, fe), cell)
    its = A.cg_solve(x, b, level, 1000, 1e-13)
    assert 0 < its < 1000
    for cell, (fv, fe) in enumerate(exact):
        gv, ge = x.download(level, cell)
        assert np.abs(gv - fv).max() < 1e-9 and np.abs(ge - fe).max() < 1e-9
    for o in (u, r, x, b, A, st):
        o.close()
