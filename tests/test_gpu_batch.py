"""GPU tests of the batched kernels (one launch for all macro-cells of a rank, inner and boundary points together):
the C-ABI entry points against the CPU oracle applied cell by cell, and the host layer with batching on against the same
host layer with batching off (per-cell kernels) on whole V-cycles."""
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


def _cells(po, level):
    """three cells of regular_octahedron_8el with their stencil tables [15][15] (rows 0..13 shares, 14 inner)"""
    import hostutil as hu

    v, c = hu.read_msh(hu.MESHES / "regular_octahedron_8el.msh")
    tabs = []
    for cv in c[[0, 3, 6]]:
        co = v[cv].reshape(12)
        tabs.append(np.vstack([po.assemble_cell_slot_stencils(co, level).reshape(14, 15), po.assemble_cell_stencil(co, level)[None, :]]))
    return np.array(tabs)


MASKS = [0x7FFF, 0x4000 | 0x2A5, 0x3FFF]


def _oracle_apply(po, dst, src, level, tab, mask, update):
    if (mask & po.MASK_INNER) and level >= 2:
        po.apply_cell(dst, src, level, tab[14], update)
    po.apply_cell_boundary(dst, src, level, tab[:14].reshape(-1), mask & po.MASK_SHELL, update)
    return dst


@pytest.mark.parametrize("level", [0, 1, 2, 3, 5])
@pytest.mark.parametrize("update", [0, 1])
def test_apply_cells_matches_the_oracle_cell_by_cell(env, level, update):
    torch, capi, host, po = env
    tabs = _cells(po, level)
    n = po.cell_size(level)
    rng = np.random.default_rng(level)
    src = [rng.standard_normal(n) for _ in range(3)]
    dst0 = [rng.standard_normal(n) for _ in range(3)]
    dsrc, ddst, dtab = [_dev(torch, a) for a in src], [_dev(torch, a) for a in dst0], _dev(torch, tabs.reshape(-1))
    capi.p1_apply_cells([t.data_ptr() for t in ddst], [t.data_ptr() for t in dsrc], level, dtab.data_ptr(), MASKS, update)
    torch.cuda.synchronize()
    for c in range(3):
        want = _oracle_apply(po, dst0[c].copy(), src[c], level, tabs[c], MASKS[c], update)
        got = ddst[c].cpu().numpy()
        assert np.abs(got - want).max() <= 1e-13 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("level", [0, 2, 4])
def test_vector_and_dot_cells(env, level):
    torch, capi, host, po = env
    import hostutil as hu

    n = po.cell_size(level)
    rng = np.random.default_rng(10 + level)
    a, b, d0 = ([rng.standard_normal(n) for _ in range(3)] for _ in range(3))
    for op, scalars in ((0, [2.0, -0.5]), (1, [0.25, 3.0]), (2, None), (3, [7.5])):
        da, db, dd = [_dev(torch, x) for x in a], [_dev(torch, x) for x in b], [_dev(torch, x) for x in d0]
        srcs = [] if op == 3 else [[t.data_ptr() for t in da], [t.data_ptr() for t in db]]
        capi.p1_vector_cells(op, [t.data_ptr() for t in dd], srcs, scalars, level, MASKS)
        torch.cuda.synchronize()
        for c in range(3):
            sel = hu.point_mask(level, MASKS[c])
            if op == 0:
                want = 2.0 * a[c] - 0.5 * b[c]
            elif op == 1:
                want = d0[c] + (0.25 * a[c] + 3.0 * b[c])
            elif op == 2:
                want = a[c] * b[c]
            else:
                want = np.full(n, 7.5)
            got = dd[c].cpu().numpy()
            assert np.array_equal(got[~sel], d0[c][~sel])
            assert np.abs(got[sel] - want[sel]).max() <= 1e-14 * max(1.0, np.abs(want).max()) if sel.any() else True
    da, db = [_dev(torch, x) for x in a], [_dev(torch, x) for x in b]
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    ws = torch.zeros(capi.dot_workspace_bytes() // 8, dtype=torch.float64, device="cuda")
    capi.p1_dot_cells([t.data_ptr() for t in da], [t.data_ptr() for t in db], level, MASKS, res.data_ptr(), ws.data_ptr())
    torch.cuda.synchronize()
    want = sum(po.dot_cell_masked(a[c], b[c], level, MASKS[c]) for c in range(3))
    assert abs(float(res[0]) - want) <= 1e-13 * max(1.0, abs(want))


@pytest.mark.parametrize("coarse_level", [0, 1, 3])
def test_grid_transfer_cells_match_the_per_cell_kernels(env, coarse_level):
    """same terms and order as the per-cell masked kernels: bit-identical"""
    torch, capi, host, po = env
    nc, nf = po.cell_size(coarse_level), po.cell_size(coarse_level + 1)
    rng = np.random.default_rng(coarse_level)
    nnc = np.array([[1, 2, 4, 1, 3, 2, 1, 2, 2, 1, 5, 4, 3, 8], [2] * 14, [1] * 14], dtype=np.float64)
    dinv = _dev(torch, (1.0 / nnc).reshape(-1))
    fine, coarse = [rng.standard_normal(nf) for _ in range(3)], [rng.standard_normal(nc) for _ in range(3)]
    # restriction
    df, dc, dc2 = [_dev(torch, x) for x in fine], [_dev(torch, x) for x in coarse], [_dev(torch, x) for x in coarse]
    capi.p1_restrict_cells([t.data_ptr() for t in dc], [t.data_ptr() for t in df], coarse_level, dinv.data_ptr(), MASKS)
    for c in range(3):
        capi.p1_restrict_cell_masked(dc2[c].data_ptr(), df[c].data_ptr(), coarse_level, nnc[c], MASKS[c])
    torch.cuda.synchronize()
    for c in range(3):
        assert np.array_equal(dc[c].cpu().numpy(), dc2[c].cpu().numpy())
    # prolongation (Replace on the masked points)
    dc = [_dev(torch, x) for x in coarse]
    df, df2 = [_dev(torch, x) for x in fine], [_dev(torch, x) for x in fine]
    capi.p1_prolongate_cells([t.data_ptr() for t in dc], [t.data_ptr() for t in df], coarse_level, dinv.data_ptr(), MASKS)
    for c in range(3):
        capi.p1_prolongate_cell_masked(dc[c].data_ptr(), df2[c].data_ptr(), coarse_level, nnc[c], MASKS[c])
    torch.cuda.synchronize()
    for c in range(3):
        assert np.array_equal(df[c].cpu().numpy(), df2[c].cpu().numpy())


def test_batched_calls_reject_bad_arguments(env):
    torch, capi, host, po = env
    a = _dev(torch, np.zeros(po.cell_size(2)))
    with pytest.raises(capi.HytegHipError):
        capi.p1_vector_cells(0, [a.data_ptr()] * 81, [[a.data_ptr()] * 81], [1.0], 2, [0x7FFF] * 81)  # > HYTEG_HIP_MAX_BATCH
    with pytest.raises(capi.HytegHipError):
        capi.p1_apply_cells([a.data_ptr()], [a.data_ptr()], 2, a.data_ptr(), [0x7FFF])  # dst aliases src


def _vcycle(host, mesh, lo, hi, smoother, batch):
    import hostutil as hu

    st = host.Storage.from_gmsh(hu.MESHES / f"{mesh}.msh")
    st.set_batch_max_level(batch)
    mo = hu.MultiCellOracle(st)
    A = host.P1ConstantOperator(st, lo, hi)
    A.compute_inverse_diagonal()
    x, b, r = (host.P1Function(st, n, lo, hi) for n in ("x", "b", "r"))
    hu.upload(x, mo.interpolate(lambda X, Y, Z: np.sin(11 * X) * np.cos(5 * Y) + Z * X, hi), hi)
    x.interpolate(0.0, hi, host.DirichletBoundary)
    hu.upload(b, mo.interpolate(lambda X, Y, Z: 1.0 + X - Y * Z, hi), hi)
    gmg = host.Solver.gmg(st, lo, hi, smoother=smoother, relax=2.0 / 3.0, pre=2, post=2, cg_max_iter=30, cg_tol=1e-12)
    gmg.solve(A, x, b, hi)
    A.apply(x, r, hi, host.Inner)
    out = hu.download(x, hi), hu.download(r, hi), r.dot(r, hi, host.Inner)
    for o in (gmg, x, b, r, A, st):
        o.close()
    return out


@pytest.mark.parametrize("mesh,lo,hi,smoother", [("regular_octahedron_8el", 0, 4, "GAUSS_SEIDEL"), ("regular_octahedron_8el", 2, 4, "JACOBI"),
                                                  ("cube_6el", 1, 3, "JACOBI"), ("pyramid_tilted_4el", 0, 3, "GAUSS_SEIDEL")])
def test_vcycle_with_batched_kernels_equals_the_per_cell_kernels(env, mesh, lo, hi, smoother):
    torch, capi, host, po = env
    xb, rb, db = _vcycle(host, mesh, lo, hi, getattr(host, smoother), 6)
    xc, rc, dc = _vcycle(host, mesh, lo, hi, getattr(host, smoother), -1)
    scale = max(np.abs(a).max() for a in xc)
    for a, b in zip(xb, xc):
        assert np.abs(a - b).max() <= 1e-11 * scale
    assert abs(db - dc) <= 1e-9 * abs(dc)


def test_gauss_seidel_sweeps_shared_between_cells_above_the_batch_level(env):
    """Above the batch level only the Gauss-Seidel / SOR sweeps (macro-cell sweep, rest, shell sweeps) share their launches
    between the cells of a rank (PrimitiveStorage::useBatchSor): a V(2,2) cycle over levels 6-7 of a 2-cell mesh with the batch
    level at 5 against the same cycle with per-cell launches throughout."""
    torch, capi, host, po = env
    xb, rb, db = _vcycle(host, "pyramid_2el", 6, 7, host.GAUSS_SEIDEL, 5)
    xc, rc, dc = _vcycle(host, "pyramid_2el", 6, 7, host.GAUSS_SEIDEL, -1)
    scale = max(np.abs(a).max() for a in xc)
    for a, b in zip(xb, xc):
        assert np.abs(a - b).max() <= 1e-11 * scale
    assert abs(db - dc) <= 1e-9 * abs(dc)


@pytest.mark.parametrize("level", [2, 3, 4, 5, 6])
@pytest.mark.parametrize("backwards", [False, True])
def test_sor_cells_match_the_per_cell_sweeps(env, level, backwards):
    """levels <= 4: same update and summation order as the per-cell plane kernel (bit-identical); level 5 is compared
    with the blocked per-cell kernel (other summation order inside an update), level 6 is the same blocked kernel batched"""
    torch, capi, host, po = env
    tabs = _cells(po, level)
    n = po.cell_size(level)
    rng = np.random.default_rng(level)
    u0, b = [rng.standard_normal(n) for _ in range(3)], [rng.standard_normal(n) for _ in range(3)]
    masks = [0x7FFF, 0x3FFF, 0x4000]  # the middle cell has no inner points selected: untouched
    du, db, dtab = [_dev(torch, a) for a in u0], [_dev(torch, a) for a in b], _dev(torch, tabs.reshape(-1))
    capi.p1_sor_cells([t.data_ptr() for t in du], [t.data_ptr() for t in db], level, dtab.data_ptr(), 1.15, masks, backwards)
    torch.cuda.synchronize()
    for c in range(3):
        got = du[c].cpu().numpy()
        if not masks[c] & po.MASK_INNER:
            assert np.array_equal(got, u0[c])
            continue
        ref = _dev(torch, u0[c])
        capi.p1_sor_cell(ref.data_ptr(), db[c].data_ptr(), level, list(tabs[c][14]), 1.15, backwards)
        torch.cuda.synchronize()
        want = ref.cpu().numpy()
        if level == 5:
            assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
        else:
            assert np.array_equal(got, want)
        oracle = po.sor_cell(u0[c].copy(), b[c], level, tabs[c][14], 1.15, backwards)
        assert np.abs(got - oracle).max() <= 1e-12 * np.abs(oracle).max()


@pytest.mark.parametrize("level", [0, 1, 2, 4, 6])
@pytest.mark.parametrize("backwards", [False, True])
def test_sor_shell_cells_match_the_per_cell_kernel(env, level, backwards):
    torch, capi, host, po = env
    import hostutil as hu

    v, c = hu.read_msh(hu.MESHES / "regular_octahedron_8el.msh")
    tables = [hu.sor_tables(v, c, level)[k] for k in (1, 4, 7)]
    n = po.cell_size(level)
    rng = np.random.default_rng(40 + level)
    u0, b, rest = ([rng.standard_normal(n) for _ in range(3)] for _ in range(3))
    masks = [0x3FFF, 0x2A5 | (0x5 << 10), 0x3C0]
    raw = np.frombuffer(capi.sor_shell_tables_bytes(tables), dtype=np.uint8).copy()
    dtab = torch.from_numpy(raw).to("cuda")
    du, db, dr = ([_dev(torch, a) for a in arrs] for arrs in (u0, b, rest))
    capi.p1_sor_shell_cells([t.data_ptr() for t in du], [t.data_ptr() for t in db], [t.data_ptr() for t in dr], level, dtab.data_ptr(), 1.1,
                            masks, backwards)
    torch.cuda.synchronize()
    for k, t in enumerate(tables):
        ref, rr = _dev(torch, u0[k]), _dev(torch, rest[k])
        capi.p1_sor_shell_cell(ref.data_ptr(), db[k].data_ptr(), rr.data_ptr(), level, t["edge_verts"], t["edge_w"], t["face_verts"],
                               t["face_w"], t["vertex_w"], 1.1, masks[k], backwards)
        torch.cuda.synchronize()
        assert np.array_equal(du[k].cpu().numpy(), ref.cpu().numpy())
