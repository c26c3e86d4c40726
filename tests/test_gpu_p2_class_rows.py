"""GPU parity of the P2 row kernel that computes every point class (p2_class_rows_kernel: inner DoFs and all boundary classes by row
waves with wave-uniform class weights), which hyteg_hip_p2_elementwise_apply_cell uses from level 3, against the CPU restatement of
P2ElementwiseOperator::gemv (oracle/p1_oracle.c ho_p2_elementwise_apply_cell), mask by mask, and against the kernels of rounds 1-2
(row kernel for the inner DoFs + thread-per-DoF boundary kernel; hyteg_hip_p2_set_class_rows_min_level( 99 )) at the levels the
oracle takes too long for."""
import numpy as np
import pytest

from conftest import REF_TET, SKEW_TET

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from hyteg_amd import capi
    from oracle import p1_oracle as po

    assert torch.cuda.is_available()
    capi.lib()
    return torch, capi, po


@pytest.fixture()
def class_rows_from_level_3(env):
    capi = env[1]
    before = capi.p2_set_class_rows_min_level(3)
    yield
    capi.p2_set_class_rows_min_level(before)


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


MASKS = tuple((0x4000 | (1 << c), 0, 1.0) for c in range(14)) + tuple(((1 << 14) | (0x3FFF & ~(1 << c)), 1, 0.5) for c in (1, 5, 8, 9, 12)) + ((0x7FFF, 0, 1.0), (1 << 14, 0, 1.0), (0x7FFF, 1, -0.5), (0x4000 | 0x2A5, 0, 2.0), (0x4000 | (1 << 6), 1, 1.0), (0x4000 | (1 << 7) | 1, 0, 1.0),
         (0x4000 | 0x3F3E, 0, 1.0))


@pytest.mark.parametrize("level", [3, 4, 5])
@pytest.mark.parametrize("tet", [REF_TET, SKEW_TET])
def test_class_rows_apply_matches_the_oracle(env, class_rows_from_level_3, level, tet):
    torch, capi, po = env
    nv, ne = po.cell_size(level), po.edge_array_size(level)
    em = po.p2_cell_element_matrices(np.asarray(tet, dtype=np.float64).reshape(12), level)
    rng = np.random.default_rng(100 + level)
    sv, se, dv0, de0 = rng.standard_normal(nv), rng.standard_normal(ne), rng.standard_normal(nv), rng.standard_normal(ne)
    dem = _dev(torch, capi.p2_build_operator_table(em))
    for mask, update, alpha in MASKS:
        wv, we = po.p2_elementwise_apply_cell(dv0.copy(), de0.copy(), sv, se, level, em, alpha, update, mask)
        dsv, dse, ddv, dde = _dev(torch, sv), _dev(torch, se), _dev(torch, dv0), _dev(torch, de0)
        capi.p2_elementwise_apply_cell(ddv.data_ptr(), dde.data_ptr(), dsv.data_ptr(), dse.data_ptr(), level, dem.data_ptr(), alpha, update, mask)
        torch.cuda.synchronize()
        gv, ge = ddv.cpu().numpy(), dde.cpu().numpy()
        scale = max(np.abs(wv).max(), np.abs(we).max(), 1.0)
        assert np.abs(gv - wv).max() <= 1e-13 * scale and np.abs(ge - we).max() <= 1e-13 * scale, (level, hex(mask), update)
        sel_v = ((mask >> po.slot_of_points(level)) & 1).astype(bool)
        sel_e = ((mask >> po.edge_classes(level)) & 1).astype(bool)
        assert np.array_equal(gv[~sel_v], dv0[~sel_v]) and np.array_equal(ge[~sel_e], de0[~sel_e]), (level, hex(mask), update)


@pytest.mark.parametrize("level", [4, 6])
def test_class_rows_restricted_to_kinds_match_the_oracle(env, class_rows_from_level_3, level):
    """the per-type applies of the P2 Gauss-Seidel smoother: some destination kinds only, with and without the inner DoFs (level 6:
    the class-rows kernel restricted to kinds; level 4: the kernels of round 2, which keep these applies below level 6)"""
    torch, capi, po = env
    nv, ne = po.cell_size(level), po.edge_array_size(level)
    em = po.p2_cell_element_matrices(np.asarray(SKEW_TET, dtype=np.float64).reshape(12), level)
    rng = np.random.default_rng(200 + level)
    sv, se, dv0, de0 = rng.standard_normal(nv), rng.standard_normal(ne), rng.standard_normal(nv), rng.standard_normal(ne)
    dem = _dev(torch, capi.p2_build_operator_table(em))
    n = 1 << level
    tet = lambda w: w * (w + 1) * (w + 2) // 6
    kind_of_edge = np.concatenate([np.full(tet(n), k) for k in range(1, 7)] + [np.full(tet(n - 1), 7)])
    assert len(kind_of_edge) == ne
    for kinds in (0x01, 0x02, 0x10, 0x80, 0xFE, 0x4D):
        for mask, update in ((0x7FFF, 0), (0x4000, 1), (0x3FFF, 0), (0x3FFF, 1)):
            # the oracle computes every kind; a kind-restricted apply leaves the other kinds' entries as they were
            fv, fe = po.p2_elementwise_apply_cell(dv0.copy(), de0.copy(), sv, se, level, em, 1.25, update, mask)
            wv = fv if kinds & 1 else dv0
            we = np.where((kinds >> kind_of_edge) & 1, fe, de0)
            dsv, dse, ddv, dde = _dev(torch, sv), _dev(torch, se), _dev(torch, dv0), _dev(torch, de0)
            capi.p2_elementwise_apply_cell(ddv.data_ptr(), dde.data_ptr(), dsv.data_ptr(), dse.data_ptr(), level, dem.data_ptr(), 1.25, update, mask,
                                           kinds=kinds)
            torch.cuda.synchronize()
            gv, ge = ddv.cpu().numpy(), dde.cpu().numpy()
            scale = max(np.abs(fv).max(), np.abs(fe).max(), 1.0)
            assert np.abs(gv - wv).max() <= 1e-13 * scale and np.abs(ge - we).max() <= 1e-13 * scale, (level, hex(kinds), hex(mask), update)
            if not kinds & 1:
                assert np.array_equal(gv, dv0)
            assert np.array_equal(ge[((kinds >> kind_of_edge) & 1) == 0], de0[((kinds >> kind_of_edge) & 1) == 0])


@pytest.mark.parametrize("level", [6, 7])
def test_class_rows_equal_the_kernels_of_round_2_to_rounding(env, level):
    """levels the oracle takes too long for: the two forms of the kernel on the same input (they differ in the order of the sum only),
    every DoF written by exactly one of the launch's two parts"""
    torch, capi, po = env
    nv, ne = capi.cell_size(level), capi.p2_edge_array_size(level)
    em = po.p2_cell_element_matrices(np.asarray(SKEW_TET, dtype=np.float64).reshape(12), level)
    dem = _dev(torch, capi.p2_build_operator_table(em))
    rng = np.random.default_rng(level)
    sv, se = _dev(torch, rng.standard_normal(nv)), _dev(torch, rng.standard_normal(ne))
    out = {}
    for first in (99, 3):
        before = capi.p2_set_class_rows_min_level(first)
        try:
            for mask, update in ((0x7FFF, 0), (0x7FFF, 1), (0x4000, 0)):
                dv, de = torch.full((nv,), 0.25, dtype=torch.float64, device="cuda"), torch.full((ne,), -0.5, dtype=torch.float64, device="cuda")
                capi.p2_elementwise_apply_cell(dv.data_ptr(), de.data_ptr(), sv.data_ptr(), se.data_ptr(), level, dem.data_ptr(), 1.0, update, mask)
                torch.cuda.synchronize()
                out[first, mask, update] = (dv.cpu().numpy(), de.cpu().numpy())
        finally:
            capi.p2_set_class_rows_min_level(before)
    for mask, update in ((0x7FFF, 0), (0x7FFF, 1), (0x4000, 0)):
        (av, ae), (bv, be) = out[99, mask, update], out[3, mask, update]
        scale = max(np.abs(av).max(), np.abs(ae).max())
        assert np.abs(av - bv).max() <= 1e-13 * scale and np.abs(ae - be).max() <= 1e-13 * scale, (level, hex(mask), update)
        # untouched entries are untouched in both
        assert np.array_equal(av == 0.25, bv == 0.25) and np.array_equal(ae == -0.5, be == -0.5)


def test_class_rows_read_rows_that_do_not_exist_as_zero(env, class_rows_from_level_3):
    """a weight that is exactly zero (neighbour outside the macro-cell) must not meet a stray value: NaNs in the DESTINATION arrays'
    neighbourhood are harmless by construction; here the source is finite and huge in the first and last entries of every kind's
    array, next to which the rows of the faces y = 0 and z = 0 are read"""
    torch, capi, po = env
    level = 4
    nv, ne = po.cell_size(level), po.edge_array_size(level)
    em = po.p2_cell_element_matrices(np.asarray(REF_TET, dtype=np.float64).reshape(12), level)
    rng = np.random.default_rng(7)
    sv, se = rng.standard_normal(nv), rng.standard_normal(ne)
    wv, we = po.p2_elementwise_apply_cell(np.zeros(nv), np.zeros(ne), sv, se, level, em, 1.0, 0, 0x7FFF)
    dem = _dev(torch, capi.p2_build_operator_table(em))
    dsv, dse = _dev(torch, sv), _dev(torch, se)
    dv, de = torch.zeros(nv, dtype=torch.float64, device="cuda"), torch.zeros(ne, dtype=torch.float64, device="cuda")
    capi.p2_elementwise_apply_cell(dv.data_ptr(), de.data_ptr(), dsv.data_ptr(), dse.data_ptr(), level, dem.data_ptr(), 1.0, 0, 0x7FFF)
    torch.cuda.synchronize()
    scale = max(np.abs(wv).max(), np.abs(we).max())
    assert np.abs(dv.cpu().numpy() - wv).max() <= 1e-13 * scale and np.abs(de.cpu().numpy() - we).max() <= 1e-13 * scale


@pytest.mark.parametrize("level", [3, 4])
@pytest.mark.parametrize("update", [0, 1])
def test_class_rows_batched_over_cells_match_the_oracle(env, level, update):
    """hyteg_hip_p2_elementwise_apply_cells_kinds with every kind: one launch of row waves for all cells, each cell with its own
    arrays, operator table and point mask (one of them without the inner DoFs, one empty)"""
    torch, capi, po = env
    from conftest import OCT_TET

    nv, ne = po.cell_size(level), po.edge_array_size(level)
    tets, masks = [REF_TET, SKEW_TET, OCT_TET, REF_TET, SKEW_TET], [0x7FFF, 0x4000 | 0x2A5, 0x3FFF, 0, 0x4000]
    rng = np.random.default_rng(31 + level)
    keep, want = [], []
    lists = {k: [] for k in ("dv", "de", "sv", "se", "tab")}
    for tet, mask in zip(tets, masks):
        em = po.p2_cell_element_matrices(np.asarray(tet, dtype=np.float64).reshape(12), level)
        sv, se, dv0, de0 = rng.standard_normal(nv), rng.standard_normal(ne), rng.standard_normal(nv), rng.standard_normal(ne)
        want.append(po.p2_elementwise_apply_cell(dv0.copy(), de0.copy(), sv, se, level, em, -1.5, update, mask))
        dev = [_dev(torch, a) for a in (dv0, de0, sv, se, capi.p2_build_operator_table(em))]
        keep.append(dev)
        for k, d in zip(("dv", "de", "sv", "se", "tab"), dev):
            lists[k].append(d.data_ptr())
    capi.p2_elementwise_apply_cells(lists["dv"], lists["de"], lists["sv"], lists["se"], level, lists["tab"], masks, -1.5, update)
    torch.cuda.synchronize()
    for (wv, we), dev in zip(want, keep):
        gv, ge = dev[0].cpu().numpy(), dev[1].cpu().numpy()
        scale = max(np.abs(wv).max(), np.abs(we).max(), 1.0)
        assert np.abs(gv - wv).max() <= 1e-13 * scale and np.abs(ge - we).max() <= 1e-13 * scale


def test_class_rows_equal_the_kernels_of_round_2_at_level_8(env):
    """the largest level of BASELINE config 4's shape class on one GPU (183 MB of DoFs per function): compared on the device"""
    torch, capi, po = env
    level = 8
    nv, ne = capi.cell_size(level), capi.p2_edge_array_size(level)
    em = po.p2_cell_element_matrices(np.asarray(SKEW_TET, dtype=np.float64).reshape(12), level)
    dem = _dev(torch, capi.p2_build_operator_table(em))
    g = torch.Generator(device="cuda").manual_seed(8)
    sv, se = torch.randn(nv, dtype=torch.float64, device="cuda", generator=g), torch.randn(ne, dtype=torch.float64, device="cuda", generator=g)
    out = {}
    for first in (99, 3):
        before = capi.p2_set_class_rows_min_level(first)
        try:
            dv, de = torch.full((nv,), 0.25, dtype=torch.float64, device="cuda"), torch.full((ne,), -0.5, dtype=torch.float64, device="cuda")
            capi.p2_elementwise_apply_cell(dv.data_ptr(), de.data_ptr(), sv.data_ptr(), se.data_ptr(), level, dem.data_ptr(), 1.0, 1, 0x7FFF)
            torch.cuda.synchronize()
            out[first] = (dv, de)
        finally:
            capi.p2_set_class_rows_min_level(before)
    (av, ae), (bv, be) = out[99], out[3]
    scale = max(float(av.abs().max()), float(ae.abs().max()))
    assert float((av - bv).abs().max()) <= 1e-13 * scale and float((ae - be).abs().max()) <= 1e-13 * scale
    assert int((bv == 0.25).sum()) == 0 and int((be == -0.5).sum()) == 0  # ADD onto every DoF: none left as it was
