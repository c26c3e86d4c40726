"""Pins of the P2 grid-transfer oracle (oracle/p2_transfer_oracle.py) by the reference's own known answers
(tests/hyteg/P2/P2QuadraticProlongation3DTest.cpp, P2QuadraticRestriction3DTest.cpp)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import p1_oracle as po  # noqa: E402
from oracle import p2_transfer_oracle as pt  # noqa: E402

SKEW_TET = ((0.1, -0.2, 0.05), (1.3, 0.1, -0.1), (0.4, 1.1, 0.2), (-0.2, 0.3, 0.9))


def _nonzeros(fv, fe):
    return int((np.abs(fv) > 1e-8).sum() + (np.abs(fe) > 1e-8).sum())


def test_support_of_a_coarse_vertex_dof():
    # P2QuadraticProlongation3DTest.cpp:46-85: unit value at vertex DoF (1,1,1), level 2 -> 1 + 2*14 + 3*24 + 24 fine DoFs
    lv = 2
    cv, ce = np.zeros(po.cell_size(lv)), np.zeros(po.edge_array_size(lv))
    cv[po.cell_index(lv, 1, 1, 1)] = 1.0
    assert _nonzeros(*pt.prolongate_cell(cv, ce, lv)) == 1 + 2 * 14 + 3 * 24 + 24


def test_support_of_a_coarse_edge_dof():
    # :87-158: unit value at the Z edge DoF (1,1,0), level 2 -> 1 + 2 + 6*2 + 6*1 + 6*1 fine DoFs
    lv = 2
    cv, ce = np.zeros(po.cell_size(lv)), np.zeros(po.edge_array_size(lv))
    ce[po.edge_index(lv, 1, 1, 0, 2)] = 1.0
    assert _nonzeros(*pt.prolongate_cell(cv, ce, lv)) == 1 + 2 + 6 * 2 + 6 * 1 + 6 * 1


@pytest.mark.parametrize("lower", [0, 1, 2, 3])
def test_prolongation_is_exact_on_quadratics(lower):
    # :160-255 (testGridTransfer3D): constants, linears and a quadratic are reproduced (squared discrete error < 1e-15)
    import hostutil as hu

    fns = [lambda p: 0.0 * p[:, 0], lambda p: 1.0 + 0.0 * p[:, 0], lambda p: 42.0 + 0.0 * p[:, 0], lambda p: 42.0 * p[:, 0],
           lambda p: 42.0 * p[:, 0] + p[:, 1] + 1337.0 * p[:, 2],
           lambda p: 2.0 * p[:, 0] ** 2 + 3.0 * p[:, 0] + 13.0 + 4.0 * p[:, 1] + 5.0 * p[:, 1] ** 2 + p[:, 2] ** 2 + 6.0]
    for fn in fns:
        cv, ce = fn(hu.cell_points(SKEW_TET, lower)), fn(po.edge_midpoints(SKEW_TET, lower))
        fv, fe = pt.prolongate_cell(cv, ce, lower)
        ev, ee = fn(hu.cell_points(SKEW_TET, lower + 1)), fn(po.edge_midpoints(SKEW_TET, lower + 1))
        scale = max(1.0, np.abs(ev).max())
        assert np.abs(fv - ev).max() <= 2e-14 * scale and np.abs(fe - ee).max() <= 2e-14 * scale


@pytest.mark.parametrize("lower", [3, 4])
def test_restriction_of_the_constant_one(lower):
    # P2QuadraticRestriction3DTest.cpp:46-88 (testWeightsInCell( 3 ), ( 4 )): at DoFs inside the macro-cell
    fv, fe = np.ones(po.cell_size(lower + 1)), np.ones(po.edge_array_size(lower + 1))
    rv, re_ = pt.restrict_cell(fv, fe, lower + 1)
    expected_v = 1.0 + 14.0 * 3.0 / 8.0 + 14.0 * (-1.0 / 8.0) + 3.0 * 24.0 * (-1.0 / 8.0) + 24.0 * (-1.0 / 8.0)
    inner_v = po.slot_of_points(lower) == 14
    assert inner_v.any() and np.abs(rv[inner_v] - expected_v).max() < 1e-13
    neighbours = {0: 6, 1: 4, 2: 6, 3: 6, 4: 4, 5: 6, 6: 4}  # X, Y, Z, XY, XZ, YZ, XYZ
    ec, cls = po.edge_coords(lower), po.edge_classes(lower)
    n = np.array([neighbours[int(o)] for o in ec[:, 3]], dtype=np.float64)
    expected_e = 1.0 + 2.0 * 3.0 / 4.0 + n * 2.0 * 0.5 + n * 0.25 + n * 0.25
    inner_e = cls == 14
    assert inner_e.any() and np.abs(re_[inner_e] - expected_e[inner_e]).max() < 1e-13


def test_neighbour_cell_scaling_of_the_restriction():
    """fine DoFs on a macro-face / -edge / -vertex shared by k cells are scaled by 1/k (restrictAdditively3D), so that the
    contributions of the k cells add up to the unscaled transpose"""
    lv = 2
    rng = np.random.default_rng(0)
    fv, fe = rng.random(po.cell_size(lv + 1)), rng.random(po.edge_array_size(lv + 1))
    nnc = np.array([2, 3, 1, 4, 2, 5, 2, 1, 2, 2, 6, 7, 3, 8], dtype=np.float64)
    a = pt.restrict_cell(fv, fe, lv + 1, nnc)
    inv = np.concatenate([1.0 / nnc, [1.0]])
    b = pt.restrict_cell(fv * inv[po.slot_of_points(lv + 1)], fe * inv[po.edge_classes(lv + 1)], lv + 1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
