"""GPU tests of the hyteg_operators seam (include/hyteg_hip.h section b-3): the entry points with the generated kernels'
argument list (arrays, twelve macro-vertex coordinates, micro_edges_per_macro_edge) against the oracle's literal restatement
of the generated micro-cell loop (ho_p1_elementwise_apply_macro_3d / ho_p2_elementwise_apply_cell), and the host-layer class
P1ElementwiseDiffusion against P1ConstantLaplaceOperator with the reference's criterion for that pair (< 1e-13,
tests/hyteg/convergence/P1JacobiConvergenceTest.cpp:117)."""
from pathlib import Path

import numpy as np
import pytest

from conftest import OCT_TET, REF_TET, SKEW_TET

pytestmark = pytest.mark.gpu
TOL = 1e-13
MESHES = Path(__file__).resolve().parent.parent / "hyteg_amd" / "data" / "meshes"


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


def _rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("tet", [REF_TET, OCT_TET, SKEW_TET])
def test_p1_apply_macro_3d_equals_the_generated_micro_cell_loop(env, level, tet):
    torch, capi, host, po = env
    rng = np.random.default_rng(31 + level)
    n = po.cell_size(level)
    src_h, d0 = rng.random(n), rng.random(n)
    src, dst = _dev(torch, src_h), _dev(torch, d0)
    capi.p1_elementwise_diffusion_apply_macro_3d(dst.data_ptr(), src.data_ptr(), tet, 1 << level)
    torch.cuda.synchronize()
    ref = po.p1_elementwise_apply_macro_3d(d0.copy(), src_h, tet, 1 << level)
    assert _rel(dst.cpu().numpy() - d0, ref - d0) < TOL  # the added part (dst itself carries d0 exactly)


def test_p1_apply_macro_3d_level8(env):
    torch, capi, host, po = env
    level, tet = 8, SKEW_TET
    rng = np.random.default_rng(8)
    n = po.cell_size(level)
    src_h, d0 = rng.random(n), rng.random(n)
    src, dst = _dev(torch, src_h), _dev(torch, d0)
    capi.p1_elementwise_diffusion_apply_macro_3d(dst.data_ptr(), src.data_ptr(), tet, 1 << level)
    torch.cuda.synchronize()
    ref = po.p1_elementwise_apply_macro_3d(d0.copy(), src_h, tet, 1 << level)
    assert _rel(dst.cpu().numpy() - d0, ref - d0) < TOL


@pytest.mark.parametrize("level", [1, 3, 5])
def test_p1_apply_macro_3d_masked_form(env, level):
    """only the selected point classes are written; Replace gives the bare operator there"""
    torch, capi, host, po = env
    tet = OCT_TET
    rng = np.random.default_rng(level)
    n = po.cell_size(level)
    src_h, d0 = rng.random(n), rng.random(n)
    src = _dev(torch, src_h)
    ref = po.p1_elementwise_apply_macro_3d(np.zeros(n), src_h, tet, 1 << level)
    slots = po.slot_of_points(level)
    for mask in (capi.MASK_INNER, 0x0240 | (1 << 10), capi.MASK_SHELL, capi.MASK_ALL, 0):
        for update in (capi.REPLACE, capi.ADD):
            dst = _dev(torch, d0)
            capi.p1_elementwise_diffusion_apply_macro_3d_masked(dst.data_ptr(), src.data_ptr(), tet, 1 << level, mask, update)
            torch.cuda.synchronize()
            got = dst.cpu().numpy()
            sel = ((mask >> slots) & 1) == 1
            assert np.array_equal(got[~sel], d0[~sel])
            if sel.any():
                assert _rel(got[sel] - (d0[sel] if update else 0.0), ref[sel]) < TOL


@pytest.mark.parametrize("level", [0, 2, 4, 7])
def test_p1_apply_macro_3d_float(env, level):
    """float instantiation (the generated operators exist for float32): tolerance 2e-6 relative L2 -- float rounding of sums of
    up to 15 terms in another order than the element-by-element scatter, FMA contraction (as tests/test_gpu_fp32.py)"""
    torch, capi, host, po = env
    rng = np.random.default_rng(level)
    n = po.cell_size(level)
    src_h, d0 = rng.random(n).astype(np.float32), rng.random(n).astype(np.float32)
    src, dst = _dev(torch, src_h), _dev(torch, d0)
    capi.p1_elementwise_diffusion_apply_macro_3d_f32(dst.data_ptr(), src.data_ptr(), OCT_TET, 1 << level)
    torch.cuda.synchronize()
    ref = po.p1_elementwise_apply_macro_3d(d0.astype(np.float64), src_h.astype(np.float64), OCT_TET, 1 << level)
    assert _rel(dst.cpu().numpy().astype(np.float64), ref) < 2e-6


@pytest.mark.parametrize("level", [0, 1, 3, 5])
def test_p1_diagonal_macro_3d(env, level):
    torch, capi, host, po = env
    n = po.cell_size(level)
    d0 = np.random.default_rng(level).random(n)
    diag = _dev(torch, d0)
    capi.p1_elementwise_diffusion_diagonal_macro_3d(diag.data_ptr(), SKEW_TET, 1 << level)
    torch.cuda.synchronize()
    ref = po.p1_elementwise_diagonal_macro_3d(d0.copy(), SKEW_TET, 1 << level)
    assert _rel(diag.cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("tet", [REF_TET, SKEW_TET])
def test_p2_apply_macro_3d_equals_the_micro_cell_loop(env, level, tet):
    torch, capi, host, po = env
    rng = np.random.default_rng(77 + level)
    nv, ne = po.cell_size(level), po.edge_array_size(level)
    sv, se, dv0, de0 = rng.random(nv), rng.random(ne), rng.random(nv), rng.random(ne)
    dv, de, dsv, dse = _dev(torch, dv0), _dev(torch, de0), _dev(torch, sv), _dev(torch, se)
    capi.p2_elementwise_diffusion_apply_macro_3d(dv.data_ptr(), de.data_ptr(), dsv.data_ptr(), dse.data_ptr(), tet, 1 << level)
    torch.cuda.synchronize()
    rv, re_ = dv0.copy(), de0.copy()
    po.p2_elementwise_apply_cell(rv, re_, sv, se, level, po.p2_cell_element_matrices(tet, level), 1.0, po.ADD, 0x7FFF)
    assert _rel(dv.cpu().numpy() - dv0, rv - dv0) < TOL
    assert _rel(de.cpu().numpy() - de0, re_ - de0) < TOL


@pytest.mark.parametrize("mesh", ["tet_1el", "pyramid_2el", "regular_octahedron_8el", "cube_6el"])
@pytest.mark.parametrize("level", [2, 4])
def test_host_elementwise_operator_equals_the_constant_operator(env, mesh, level):
    """P1JacobiConvergenceTest.cpp:100-117: apply and three Jacobi sweeps of the elementwise and the constant-stencil operator
    on the same data differ by less than 1e-13 (there: max-norm of the difference after each sweep)"""
    torch, capi, host, po = env
    st = host.Storage.from_gmsh(MESHES / f"{mesh}.msh")
    st.set_stream(torch.cuda.current_stream().cuda_stream)
    const = host.P1ConstantOperator(st, level, level)
    elem = host.P1ElementwiseDiffusion(st, level, level)
    const.compute_inverse_diagonal()
    elem.compute_inverse_diagonal()
    n = capi.cell_size(level)
    rng = np.random.default_rng(5)
    fs = {k: host.P1Function(st, k, level, level) for k in ("src", "rhs", "a", "b", "a2", "b2")}
    for k in ("src", "rhs", "a", "b"):
        for c in range(st.n_local_cells):
            fs[k].upload_cell(c, level, rng.random(n))
        fs[k].sync_shared(level, host.All)
    fs["a2"].assign([1.0], [fs["a"]], level, host.All)
    fs["b2"].assign([1.0], [fs["b"]], level, host.All)

    def maxdiff(f, g):
        return max(np.abs(f.download_cell(c, level) - g.download_cell(c, level)).max() for c in range(st.n_local_cells))

    # inverse diagonals
    assert maxdiff(const.inverse_diagonal(level, level), elem.inverse_diagonal()) < 1e-12  # entries are O(2^level)
    for flag in (host.Inner, host.All):
        for update in (host.Replace, host.Add):
            fs["a"].assign([1.0], [fs["a2"]], level, host.All)
            fs["b"].assign([1.0], [fs["a2"]], level, host.All)
            const.apply(fs["src"], fs["a"], level, flag, update)
            elem.apply(fs["src"], fs["b"], level, flag, update)
            assert maxdiff(fs["a"], fs["b"]) < 1e-13
    # Jacobi sweeps, alternating between two functions as the reference's loop does
    x_c, y_c, x_e, y_e = fs["a"], fs["a2"], fs["b"], fs["b2"]
    x_e.assign([1.0], [x_c], level, host.All)
    y_c.assign([1.0], [x_c], level, host.All)
    y_e.assign([1.0], [x_c], level, host.All)
    for sweep in range(3):
        const.smooth_jac(y_c, fs["rhs"], x_c, 0.6, level, host.Inner)
        elem.smooth_jac(y_e, fs["rhs"], x_e, 0.6, level, host.Inner)
        assert maxdiff(y_c, y_e) < 1e-13
        x_c, y_c, x_e, y_e = y_c, x_c, y_e, x_e
    for f in fs.values():
        f.close()
    const.close()
    elem.close()
    st.close()
