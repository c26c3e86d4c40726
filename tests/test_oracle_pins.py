"""Pins the CPU oracle (oracle/p1_oracle.c) to the reference's own known-answer tests and
properties for the P1 hot path.  CPU only."""
import numpy as np
import pytest

from oracle import p1_oracle as po
from conftest import OCT_TET, REF_TET, SKEW_TET


# tests/hyteg/Indexing/CommonIndexingTest.cpp:166-175  (macroCellSize(width))
def test_macro_cell_size_known_answers():
    expect = {1: 1, 2: 4, 3: 10, 4: 20, 5: 35, 6: 56, 7: 84, 8: 120, 9: 165, 10: 220}
    for w, s in expect.items():
        assert (w * (w + 1) * (w + 2)) // 6 == s
    # levels: width = 2^L + 1
    assert po.cell_size(0) == 4 and po.cell_size(1) == 10 and po.cell_size(2) == 35
    assert po.cell_size(5) == 6545 and po.cell_size(7) == 366145 and po.cell_size(8) == 2862209
    assert po.cell_inner_size(5) == 4495 and po.cell_inner_size(8) == 2731135 and po.cell_inner_size(7) == 333375


# tests/hyteg/Indexing/VertexDoFMacroCellIndexingTest.cpp:61-127
_L2 = dict(C=20, E=21, W=19, N=23, S=16, NW=22, SE=17, TC=29, TW=28, TS=26, TSE=27, BC=6, BE=7, BN=10, BNW=9)
_L3 = dict(C=54, E=55, W=53, N=61, S=46, NW=60, SE=47, TC=89, TW=88, TS=82, TSE=83, BC=10, BE=11, BN=18, BNW=17)
_L3TOP = dict(C=150, E=151, W=149, N=153, S=146, NW=152, SE=147, TC=159, TW=158, TS=156, TSE=157, BC=136, BE=137,
              BN=140, BNW=139)


@pytest.mark.parametrize("level,centre,table", [(2, (1, 1, 1), _L2), (3, (1, 1, 1), _L3), (3, (1, 1, 5), _L3TOP)])
def test_stencil_neighbour_index_tables(level, centre, table):
    for name, off in zip(po.STENCIL_NAMES, po.STENCIL_OFFSETS):
        x, y, z = (c + o for c, o in zip(centre, off))
        assert po.cell_index(level, x, y, z) == table[name], name


def test_layout_is_a_bijection_in_memory_order():
    for level in (0, 1, 2, 3, 4):
        c = po.cell_coords(level)
        idx = [po.cell_index(level, *map(int, p)) for p in c]
        assert idx == list(range(po.cell_size(level)))


# tests/hyteg/Indexing/VertexDoFMacroCellIndexingTest.cpp:129-135 (isOnCellFace counts)
def test_on_cell_primitive_classification():
    # slots: edges 0-5, faces 6-9, vertices 10-13, interior -1
    assert po.prim_slot(2, 0, 0, 0) == 10 and po.prim_slot(3, 0, 0, 0) == 10
    assert po.prim_slot(3, 8, 0, 0) == 11          # vertex 1
    assert po.prim_slot(3, 7, 0, 0) == 0           # on 2 faces -> edge 0
    assert po.prim_slot(3, 0, 7, 0) == 1           # edge 1
    assert po.prim_slot(3, 0, 8, 0) == 12 and po.prim_slot(3, 0, 0, 8) == 13
    assert po.prim_slot(3, 1, 7, 0) == 2 and po.prim_slot(3, 0, 0, 3) == 3
    assert po.prim_slot(3, 3, 0, 5) == 4 and po.prim_slot(3, 0, 3, 5) == 5
    assert po.prim_slot(3, 2, 2, 0) == 6 and po.prim_slot(3, 2, 0, 2) == 7
    assert po.prim_slot(3, 0, 2, 2) == 8 and po.prim_slot(3, 2, 2, 4) == 9
    assert po.prim_slot(3, 1, 1, 1) == -1


# (tolerance 2e-14 relative: the generated code uses 16-digit decimal literals such as 0.1666666666666667)
# ---- element matrix vs the reference's generated FEniCS code, compiled in place (oracle/_ref) ----
def _ref_or_skip():
    ref = po.ref_fenics()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference not mounted)")
    return ref


@pytest.mark.parametrize("tet", [REF_TET, OCT_TET, SKEW_TET])
def test_element_matrices_match_reference_fenics(tet):
    ref = _ref_or_skip()
    c = np.asarray(tet, dtype=np.float64).reshape(12)
    A = np.empty(16)
    ref.ref_p1_tet_diffusion(po._p(A), po._p(c))
    K = po.p1_tet_diffusion(c)
    assert np.allclose(K, A.reshape(4, 4), rtol=0, atol=2e-14 * np.abs(A).max())
    ref.ref_p1_tet_mass(po._p(A), po._p(c))
    M = po.p1_tet_mass(c)
    assert np.allclose(M, A.reshape(4, 4), rtol=0, atol=2e-14 * np.abs(A).max())


@pytest.mark.parametrize("tet", [REF_TET, OCT_TET, SKEW_TET])
@pytest.mark.parametrize("form", range(2, 9))
def test_stokes_block_element_matrices_match_reference_fenics(tet, form):
    """div x/y/z, divT x/y/z, PSPG (the blocks of src/mixed_operator/P1P1StokesOperator.hpp:51-64) against the reference's
    generated p1_tet_div_tet.h / p1_tet_divt_tet.h / p1_tet_pspg_tet.h compiled in place; these matrices are NOT
    symmetric, so this also pins the index convention A[test][trial] (row-major, P1FenicsForm.hpp:96-124 reads row 0)"""
    ref = _ref_or_skip()
    A = po.ref_element_matrix(ref, tet, form)
    mine = po.element_matrix(tet, form)
    assert np.abs(A).max() > 0
    assert np.allclose(mine, A, rtol=0, atol=2e-14 * np.abs(A).max())
    if 2 <= form <= 7:  # div_k is the transpose of divT_k
        other = po.element_matrix(tet, form + 3 if form <= 4 else form - 3)
        assert np.array_equal(mine, other.T)


def test_unit_tet_element_row_known_answer():
    # reference header on the unit reference tet returns row0 = [0.5,-1/6,-1/6,-1/6]
    K = po.p1_tet_diffusion(REF_TET)
    assert np.allclose(K[0], [0.5, -1 / 6, -1 / 6, -1 / 6], atol=1e-15)


# tests/hyteg/vertexdofspace/VertexDoFStencilAssemblyTest.cpp:79,85: row sum 0; weights halve per level
@pytest.mark.parametrize("tet", [REF_TET, OCT_TET, SKEW_TET])
def test_stencil_rowsum_zero_and_halving(tet):
    prev = None
    for level in range(2, 13):
        w = po.assemble_cell_stencil(tet, level)
        assert abs(w.sum()) < 1e-13
        assert w[7] > 0
        if prev is not None:
            assert np.allclose(w, 0.5 * prev, rtol=1e-11, atol=1e-14)
        prev = w


def test_stencil_symmetry():
    # constant-coefficient operator on an affine cell: w(d) == w(-d)
    for tet in (REF_TET, OCT_TET, SKEW_TET):
        w = po.assemble_cell_stencil(tet, 4)
        for k, off in enumerate(po.STENCIL_OFFSETS):
            j = po.STENCIL_OFFSETS.index(tuple(-o for o in off))
            assert abs(w[k] - w[j]) < 1e-13


# The 24 micro-tets around an inner vertex as (dx,dy,dz) tuples, transcribed as DATA from
# src/hyteg/p1functionspace/P1Elements.hpp:93-143 (white/blue/green up/down cells); used to assemble the
# stencil in Python from the reference's OWN element matrices (oracle/_ref) and compare with the oracle.
_SD = dict(zip(po.STENCIL_NAMES, po.STENCIL_OFFSETS))
_MICRO_TETS = [
    "C BC BE BN", "C S SE TS", "C W NW TW", "C N E TC",        # white up
    "C W BC S", "C E SE BE", "C N NW BN", "C TS TC TW",        # white down
    "C BC BN BNW", "C W S TS", "C E SE TSE", "C NW N TC",      # blue up
    "C BC S SE", "C W NW BNW", "C E BN N", "C TC TS TSE",      # blue down
    "C W BC BNW", "C E BE BN", "C TC TW NW", "C SE TS TSE",    # green up
    "C BC BE SE", "C BN BNW NW", "C E TSE TC", "C W TS TW",    # green down
]


@pytest.mark.parametrize("tet", [REF_TET, OCT_TET, SKEW_TET])
@pytest.mark.parametrize("level", [2, 5, 8])
@pytest.mark.parametrize("form", [0, 2, 3, 4, 5, 6, 7, 8])
def test_cell_stencil_matches_assembly_from_reference_element_matrices(tet, level, form):
    ref = _ref_or_skip()
    w = dict.fromkeys(po.STENCIL_NAMES, 0.0)
    for cellspec in _MICRO_TETS:
        names = cellspec.split()
        coords = np.concatenate([po.coordinate_from_index(tet, level, *(1 + o for o in _SD[n])) for n in names])
        A = po.ref_element_matrix(ref, coords, form)
        for j, n in enumerate(names):
            w[n] += A[0, j]  # row 0 of the row-major hyteg::Matrix: test function of the centre vertex (P1FenicsForm.hpp:96-124)
    mine = po.assemble_cell_stencil(tet, level, form)
    expect = np.array([w[n] for n in po.STENCIL_NAMES])
    assert np.allclose(mine, expect, rtol=0, atol=1e-13 * np.abs(expect).max())


# tests/hyteg/P1/P1LaplaceOperator3DTest.cpp:48,109-126: A u = 0 for u in {0,1,42x,42x+y+1337z}, < 2.8e-13
@pytest.mark.parametrize("tet", [REF_TET, OCT_TET, SKEW_TET])
@pytest.mark.parametrize("level", [2, 3, 4])
def test_laplace_annihilates_constants_and_linears(tet, level):
    w = po.assemble_cell_stencil(tet, level)
    fns = [lambda x, y, z: 0 * x, lambda x, y, z: 0 * x + 1.0, lambda x, y, z: 42 * x,
           lambda x, y, z: 42 * x + y + 1337 * z]
    npts = po.cell_inner_size(level)
    for fn in fns:
        u = po.interpolate(tet, level, fn)
        r = np.zeros_like(u)
        po.apply_cell(r, u, level, w)
        err = np.sqrt(po.dot(r, r, level) / npts)
        assert err < 2.8e-13


def test_apply_add_is_replace_plus_old():
    rng = np.random.default_rng(1)
    level = 3
    w = po.assemble_cell_stencil(SKEW_TET, level)
    u = rng.random(po.cell_size(level))
    d0 = rng.random(po.cell_size(level))
    r = np.zeros_like(u)
    po.apply_cell(r, u, level, w, po.REPLACE)
    d = d0.copy()
    po.apply_cell(d, u, level, w, po.ADD)
    m = po.inner_mask(level)
    assert np.array_equal(d[~m], d0[~m])            # boundary untouched
    assert np.array_equal(d[m], r[m] + d0[m])       # old value added last (add.cpp)
    assert np.all(r[~m] == 0)


def test_apply_matches_dense_numpy_restatement():
    """Independent check of the loop nest: build the operator as explicit index arithmetic in numpy."""
    rng = np.random.default_rng(7)
    for level in (2, 3, 4):
        w = rng.standard_normal(15)
        u = rng.random(po.cell_size(level))
        r = np.zeros_like(u)
        po.apply_cell(r, u, level, w)
        c = po.cell_coords(level)
        m = po.inner_mask(level)
        lut = {tuple(p): i for i, p in enumerate(map(tuple, c))}
        ref = np.zeros_like(u)
        for i in np.nonzero(m)[0]:
            x, y, z = c[i]
            ref[i] = sum(w[k] * u[lut[(x + o[0], y + o[1], z + o[2])]] for k, o in enumerate(po.STENCIL_OFFSETS))
        assert np.allclose(r, ref, rtol=1e-13, atol=1e-13)


def test_gs_equals_sor_relax_one_and_fixed_point():
    rng = np.random.default_rng(3)
    level = 3
    w = po.assemble_cell_stencil(OCT_TET, level)
    rhs = rng.random(po.cell_size(level))
    u0 = rng.random(po.cell_size(level))
    a, b = u0.copy(), u0.copy()
    po.gs_cell(a, rhs, level, w)
    po.sor_cell(b, rhs, level, w, 1.0)
    assert np.allclose(a, b, rtol=1e-14, atol=1e-15)
    # exact solution is a fixed point: choose u*, rhs = A u* on the interior
    ustar = rng.random(po.cell_size(level))
    r = np.zeros_like(ustar)
    po.apply_cell(r, ustar, level, w)
    for bw in (False, True):
        u = ustar.copy()
        po.sor_cell(u, r, level, w, 0.7, backwards=bw)
        assert np.allclose(u, ustar, rtol=1e-12, atol=1e-13)


def test_sor_backwards_is_mirror_order():
    """Sequential semantics: forward sweep uses updated W,S,SE,BC,BE,BN,BNW; backward the opposite set."""
    rng = np.random.default_rng(5)
    level = 2
    w = rng.standard_normal(15)
    w[7] = 10.0
    rhs = rng.random(po.cell_size(level))
    u0 = rng.random(po.cell_size(level))
    for backwards in (False, True):
        u = u0.copy()
        po.sor_cell(u, rhs, level, w, 1.3, backwards=backwards)
        # python restatement with explicit ordering
        c = po.cell_coords(level)
        lut = {tuple(p): i for i, p in enumerate(map(tuple, c))}
        order = [i for i in range(len(c)) if po.inner_mask(level)[i]]
        if backwards:
            order = order[::-1]
        v = u0.copy()
        for i in order:
            x, y, z = c[i]
            s = rhs[i] - sum(w[k] * v[lut[(x + o[0], y + o[1], z + o[2])]] for k, o in enumerate(po.STENCIL_OFFSETS)
                             if k != 7)
            v[i] = 1.3 * s / w[7] + (1 - 1.3) * v[i]
        assert np.allclose(u, v, rtol=1e-13, atol=1e-14)


def test_jacobi_composition():
    rng = np.random.default_rng(11)
    level = 3
    w = po.assemble_cell_stencil(SKEW_TET, level)
    src = rng.random(po.cell_size(level))
    rhs = rng.random(po.cell_size(level))
    dst = np.zeros_like(src)
    po.jacobi_cell(dst, rhs, src, level, w, 2.0 / 3.0)
    t = np.zeros_like(src)
    po.apply_cell(t, src, level, w)
    m = po.inner_mask(level)
    expect = src[m] + (2.0 / 3.0) * ((rhs[m] - t[m]) / w[7])
    assert np.allclose(dst[m], expect, rtol=1e-14, atol=1e-15)
    invd = np.full_like(src, 1.0 / w[7])
    d2 = np.zeros_like(src)
    po.jacobi_cell(d2, rhs, src, level, w, 2.0 / 3.0, invdiag=invd)
    assert np.array_equal(d2, dst)


# tests/hyteg/vertexdofspace/VertexDoFLinearProlongation3DTest.cpp:49,107-137: prolongation reproduces
# constants and linears (squared error < 1e-15) -- single macro-cell: all nnc = 1
@pytest.mark.parametrize("tet", [REF_TET, OCT_TET])
@pytest.mark.parametrize("lower", [0, 1, 2, 3])
def test_prolongation_exact_on_linears(tet, lower):
    ones = np.ones(14)
    fns = [lambda x, y, z: 0 * x, lambda x, y, z: 0 * x + 1.0, lambda x, y, z: 0 * x + 42.0, lambda x, y, z: 42 * x,
           lambda x, y, z: 42 * x + y]
    for fn in fns:
        uc = po.interpolate(tet, lower, fn)
        uf = np.full(po.cell_size(lower + 1), 123.0)
        po.prolongate_prepare(uf, lower + 1, po.REPLACE)
        po.prolongate_cell(uc, uf, lower, ones)
        exact = po.interpolate(tet, lower + 1, fn)
        assert float(((uf - exact) ** 2).sum()) < 1e-15 * max(1.0, float((exact ** 2).max()))


def test_restriction_is_scaled_transpose_of_prolongation():
    """With all nnc = 1, R = P^T (full weighting): <R f, c> == <f, P c> for random f, c."""
    rng = np.random.default_rng(2)
    ones = np.ones(14)
    for lc in (1, 2, 3):
        f = rng.random(po.cell_size(lc + 1))
        c = rng.random(po.cell_size(lc))
        Rf = np.zeros(po.cell_size(lc))
        po.restrict_cell(Rf, f, lc, ones)
        Pc = np.zeros(po.cell_size(lc + 1))
        po.prolongate_cell(c, Pc, lc, ones)
        assert abs(Rf @ c - f @ Pc) < 1e-11 * abs(f @ Pc)


def test_grid_transfer_neighbour_cell_scaling():
    """The 1/numNeighborCells factors: summing the per-cell partial results of `nnc` identical
    cells must reproduce the unscaled single-cell result on every shared primitive
    (P1toP1LinearRestriction.cpp:343-345 additive communication)."""
    rng = np.random.default_rng(4)
    lc = 2
    nnc = np.array([3, 4, 5, 6, 7, 8, 2, 2, 2, 2, 9, 10, 11, 12], dtype=np.float64)
    ones = np.ones(14)
    f = rng.random(po.cell_size(lc + 1))
    # make the fine function "consistent": scaling applies to the fine point's primitive
    R1 = np.zeros(po.cell_size(lc))
    po.restrict_cell(R1, f, lc, ones)
    Rn = np.zeros(po.cell_size(lc))
    po.restrict_cell(Rn, f, lc, nnc)
    # reconstruct: scale fine values by 1/nnc(prim) then restrict with ones
    cf = po.cell_coords(lc + 1)
    scale = np.array([1.0 if po.prim_slot(lc + 1, *map(int, p)) < 0 else 1.0 / nnc[po.prim_slot(lc + 1, *map(int, p))]
                      for p in cf])
    R2 = np.zeros(po.cell_size(lc))
    po.restrict_cell(R2, f * scale, lc, ones)
    assert np.allclose(Rn, R2, rtol=1e-14, atol=1e-15)
    # prolongation: target-side scaling
    c = rng.random(po.cell_size(lc))
    P1 = np.zeros(po.cell_size(lc + 1))
    po.prolongate_cell(c, P1, lc, ones)
    Pn = np.zeros(po.cell_size(lc + 1))
    po.prolongate_cell(c, Pn, lc, nnc)
    assert np.allclose(Pn, P1 * scale, rtol=1e-14, atol=1e-15)


def test_prolongate_prepare_add_zeroes_only_the_shell():
    level = 3
    a = np.full(po.cell_size(level), 5.0)
    po.prolongate_prepare(a, level, po.ADD)
    c = po.cell_coords(level)
    on_boundary = np.array([po.prim_slot(level, *map(int, p)) >= 0 for p in c])
    assert np.all(a[on_boundary] == 0) and np.all(a[~on_boundary] == 5.0)
    po.prolongate_prepare(a, level, po.REPLACE)
    assert np.all(a == 0)


def test_vector_kernels_touch_interior_only():
    rng = np.random.default_rng(9)
    level = 3
    n = po.cell_size(level)
    a, b, c3 = rng.random(n), rng.random(n), rng.random(n)
    m = po.inner_mask(level)
    d = np.full(n, -1.0)
    po.assign(d, [2.0, -3.0, 0.5], [a, b, c3], level)
    assert np.allclose(d[m], 2 * a[m] - 3 * b[m] + 0.5 * c3[m], rtol=1e-15) and np.all(d[~m] == -1)
    d = np.full(n, -1.0)
    po.add(d, [2.0], [a], level)
    assert np.allclose(d[m], -1 + 2 * a[m]) and np.all(d[~m] == -1)
    d = np.full(n, -1.0)
    po.mult_elementwise(d, [a, b], level)
    assert np.allclose(d[m], a[m] * b[m]) and np.all(d[~m] == -1)
    assert abs(po.dot(a, b, level) - float(a[m] @ b[m])) < 1e-12 * float(a[m] @ b[m])
