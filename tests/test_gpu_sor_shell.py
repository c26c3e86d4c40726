"""GPU tests of Gauss-Seidel / SOR on shared macro-faces, -edges and -vertices: the C-ABI kernel against the CPU
oracle (ho_sor_shell_cell), the host layer's multi-cell smooth_sor against the global-matrix restatement of the
reference's schedule, and the reference's own known answer: P1GMG3DConvergenceTest.cpp:52-146 (V(3,3) with
Gauss-Seidel on regular_octahedron_8el, levels 0..3, squared residual ratio < 3.2e-2 in each of 4 cycles)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from hyteg_amd import capi, host
    from oracle import p1_oracle as po

    assert torch.cuda.is_available()
    capi.lib()
    host.lib()
    return torch, capi, host, po


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 6, 7, 8])
@pytest.mark.parametrize("backwards", [False, True])
def test_sor_shell_cell_matches_the_oracle(env, level, backwards):
    torch, capi, host, po = env
    import hostutil as hu

    v, c = hu.read_msh(hu.MESHES / "regular_octahedron_8el.msh")
    cell = (level + (3 if backwards else 0)) % len(c)
    t = hu.sor_tables(v, c, level)[cell]
    rng = np.random.default_rng(100 + level)
    n = po.cell_size(level)
    u0, b, rest = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    for mask, relax in ((po.MASK_SHELL, 1.0), (0x2A5 | (0x5 << 10), 1.25), (0xF << 6, 0.8), (0x3F, 1.0), (0xF << 10, 1.1)):
        want = po.sor_shell_cell(u0.copy(), b, rest.copy(), level, t["edge_verts"], t["edge_w"], t["face_verts"], t["face_w"],
                                 t["vertex_w"], relax, mask, backwards)
        du, db, dr = _dev(torch, u0), _dev(torch, b), _dev(torch, rest)
        capi.p1_sor_shell_cell(du.data_ptr(), db.data_ptr(), dr.data_ptr(), level, t["edge_verts"], t["edge_w"], t["face_verts"],
                               t["face_w"], t["vertex_w"], relax, mask, backwards)
        torch.cuda.synchronize()
        got = du.cpu().numpy()
        sel = hu.point_mask(level, mask)
        assert np.array_equal(got[~sel], u0[~sel])            # nothing else is touched
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 1e-12 * scale, (level, mask, np.abs(got - want).max() / scale)


def test_sor_shell_cell_level_10_two_rows_per_thread(env):
    """1022 face rows: the sweep kernel's two-rows-per-thread instantiation; oracle on the same data"""
    torch, capi, host, po = env
    import hostutil as hu

    level = 10
    v, c = hu.read_msh(hu.MESHES / "regular_octahedron_8el.msh")
    t = hu.sor_tables(v, c, level)[5]
    rng = np.random.default_rng(5)
    n = po.cell_size(level)
    u0, b, rest = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    for backwards in (False, True):
        want = po.sor_shell_cell(u0.copy(), b, rest.copy(), level, t["edge_verts"], t["edge_w"], t["face_verts"], t["face_w"],
                                 t["vertex_w"], 1.0, po.MASK_SHELL, backwards)
        du, db, dr = _dev(torch, u0), _dev(torch, b), _dev(torch, rest)
        capi.p1_sor_shell_cell(du.data_ptr(), db.data_ptr(), dr.data_ptr(), level, t["edge_verts"], t["edge_w"], t["face_verts"],
                               t["face_w"], t["vertex_w"], 1.0, po.MASK_SHELL, backwards)
        torch.cuda.synchronize()
        got = du.cpu().numpy()
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
        del du, db, dr


def test_sor_shell_cell_rejects_bad_descriptors(env):
    torch, capi, host, po = env
    import hostutil as hu

    v, c = hu.read_msh(hu.MESHES / "regular_octahedron_8el.msh")
    t = hu.sor_tables(v, c, 2)[0]
    n = po.cell_size(2)
    a, b, r = (_dev(torch, np.zeros(n)) for _ in range(3))
    bad = [list(p) for p in t["edge_verts"]]
    bad[2] = [0, 3]  # edge 2 joins vertices 1 and 2
    with pytest.raises(capi.HytegHipError):
        capi.p1_sor_shell_cell(a.data_ptr(), b.data_ptr(), r.data_ptr(), 2, bad, t["edge_w"], t["face_verts"], t["face_w"], t["vertex_w"],
                               1.0, po.MASK_SHELL)
    with pytest.raises(capi.HytegHipError):
        capi.p1_sor_shell_cell(a.data_ptr(), a.data_ptr(), r.data_ptr(), 2, t["edge_verts"], t["edge_w"], t["face_verts"], t["face_w"],
                               t["vertex_w"], 1.0, po.MASK_SHELL)


@pytest.mark.parametrize("mesh,level", [("regular_octahedron_8el", 2), ("regular_octahedron_8el", 3), ("cube_6el", 3),
                                        ("pyramid_tilted_4el", 4), ("pyramid_2el", 1)])
@pytest.mark.parametrize("backwards", [False, True])
@pytest.mark.parametrize("batch", [6, -1])
def test_host_smooth_sor_matches_the_global_schedule(env, mesh, level, backwards, batch):
    torch, capi, host, po = env
    import hostutil as hu

    v, c = hu.read_msh(hu.MESHES / f"{mesh}.msh")
    glob = hu.GlobalSweepOracle(v, c, level)
    rng = np.random.default_rng(level)
    u, b = rng.standard_normal(glob.ndof), rng.standard_normal(glob.ndof)
    st = host.Storage.from_gmsh(hu.MESHES / f"{mesh}.msh")
    st.set_batch_max_level(batch)  # 6: one launch for all cells (p1_batch.hip); -1: one launch per cell and kernel
    A = host.P1ConstantOperator(st, level, level)
    x, rhs = host.P1Function(st, "x", level, level), host.P1Function(st, "b", level, level)
    for relax in (1.0, 1.2):
        hu.upload(x, glob.to_cells(u), level)
        hu.upload(rhs, glob.to_cells(b), level)
        A.smooth_sor(x, rhs, relax, level, host.Inner, backwards)
        want = glob.sweep(u, b, relax, backwards)
        got = hu.download(x, level)
        for g, a in zip(glob.gidx, got):
            assert np.abs(a - want[g]).max() <= 1e-12 * np.abs(want).max()
        # all copies of a shared DoF carry the same bits
        ref = glob.to_global(got)
        for g, a in zip(glob.gidx, got):
            assert np.array_equal(a, ref[g])
    for o in (x, rhs, A, st):
        o.close()


def test_host_smooth_sor_level_6_matches_the_cell_centric_oracle(env):
    """bigger level (edges longer than one wave, many face rows) against the oracle's C kernels composed the same way"""
    torch, capi, host, po = env
    import hostutil as hu

    mesh, level = "regular_octahedron_8el", 6
    v, c = hu.read_msh(hu.MESHES / f"{mesh}.msh")
    st = host.Storage.from_gmsh(hu.MESHES / f"{mesh}.msh")
    mo = hu.MultiCellOracle(st)
    u = mo.interpolate(lambda X, Y, Z: np.sin(9 * X) * np.cos(7 * Y) + Z * X, level)
    b = mo.interpolate(lambda X, Y, Z: np.cos(5 * X + Y) - Z, level)
    masks = hu.dirichlet_masks(v, c)
    assert masks == [st.mask(i, host.Inner) for i in range(len(c))]

    A = host.P1ConstantOperator(st, level, level)
    x, rhs = host.P1Function(st, "x", level, level), host.P1Function(st, "b", level, level)
    for backwards in (False, True):
        hu.upload(x, u, level)
        hu.upload(rhs, b, level)
        A.smooth_sor(x, rhs, 1.0, level, host.Inner, backwards)
        got = hu.download(x, level)
        sweep = hu.CellCentricSweep(v, c, level)
        want = sweep.sweep_with(lambda arrs: mo.sum_shared(arrs, level, host.Inner), u, b, masks, 1.0, backwards)
        for a, w in zip(got, want):
            assert np.abs(a - w).max() <= 1e-12 * max(np.abs(w).max() for w in want)
    for o in (x, rhs, A, st):
        o.close()


def test_gmg_v33_gauss_seidel_on_the_octahedron_levels_0_to_3(env):
    """tests/hyteg/convergence/P1GMG3DConvergenceTest.cpp: the reference's own known answer for this path."""
    torch, capi, host, po = env
    import hostutil as hu

    st = host.Storage.from_gmsh(hu.MESHES / "regular_octahedron_8el.msh")
    mo = hu.MultiCellOracle(st)
    lo, hi = 0, 3
    A = host.P1ConstantOperator(st, lo, hi)
    u, f, r, one = (host.P1Function(st, n, lo, hi) for n in ("u", "f", "r", "one"))
    rng = np.random.default_rng(1)
    exact = mo.interpolate(lambda X, Y, Z: np.sin(X) * np.sinh(Y) * Z, hi)
    rand = mo.interpolate(lambda X, Y, Z: rng.random(X.shape), hi)  # copies made consistent by interpolate()'s sync
    init = []
    for i, (e, q) in enumerate(zip(exact, rand)):
        inner = hu.point_mask(hi, st.mask(i, host.Inner))
        init.append(np.where(inner, q, e))
    hu.upload(u, init, hi)
    one.interpolate(1.0, hi, host.All)
    npoints = one.dot(one, hi, host.Inner)
    gmg = host.Solver.gmg(st, lo, hi, smoother=host.GAUSS_SEIDEL, relax=1.0, pre=3, post=3)

    def res2():
        A.apply(u, r, hi, host.Inner)
        return r.dot(r, hi, host.Inner) / npoints

    last = res2()
    for cycle in range(4):
        gmg.solve(A, u, f, hi)
        now = res2()
        assert now / last < 3.2e-2, (cycle, now / last)
        last = now
    for o in (gmg, u, f, r, one, A, st):
        o.close()


def test_gmg_v33_gauss_seidel_on_the_unit_cube_of_six_tetrahedra(env):
    """the MultigridStudies cube is meshCuboid(1,1,1) = 6 tetrahedra (data/meshes/3D/cube_6el.msh); same
    cycle and the same bound as P1GMG3DConvergenceTest (the reference has no pinned number for this mesh; measured ~1e-2)."""
    torch, capi, host, po = env
    import hostutil as hu

    st = host.Storage.from_gmsh(hu.MESHES / "cube_6el.msh")
    mo = hu.MultiCellOracle(st)
    lo, hi = 0, 4
    A = host.P1ConstantOperator(st, lo, hi)
    u, f, r, one = (host.P1Function(st, n, lo, hi) for n in ("u", "f", "r", "one"))
    rng = np.random.default_rng(2)
    exact = mo.interpolate(lambda X, Y, Z: np.sin(X) * np.sinh(Y) * Z, hi)
    rand = mo.interpolate(lambda X, Y, Z: rng.random(X.shape), hi)
    hu.upload(u, [np.where(hu.point_mask(hi, st.mask(i, host.Inner)), q, e) for i, (e, q) in enumerate(zip(exact, rand))], hi)
    one.interpolate(1.0, hi, host.All)
    npoints = one.dot(one, hi, host.Inner)
    gmg = host.Solver.gmg(st, lo, hi, smoother=host.GAUSS_SEIDEL, relax=1.0, pre=3, post=3)

    def res2():
        A.apply(u, r, hi, host.Inner)
        return r.dot(r, hi, host.Inner) / npoints

    last = res2()
    for cycle in range(4):
        gmg.solve(A, u, f, hi)
        now = res2()
        assert now / last < 3.2e-2, (cycle, now / last)
        last = now
    for o in (gmg, u, f, r, one, A, st):
        o.close()
