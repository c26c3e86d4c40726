"""CPU checks of the multi-cell SOR / Gauss-Seidel restatement (no GPU): the cell-centric composition used by the
host layer (partial sums `rest`, total weights, redundant sweeps on every copy; oracle/p1_oracle.c ho_sor_shell_cell)
against a global-matrix restatement of the reference's schedule (tests/hostutil.py GlobalSweepOracle; P1Operator.hpp:348-418,
908-1007, 1352-1503).  Parity with the reference itself is pinned through its convergence tests, see
tests/test_gpu_host.py (P1GMG3DConvergenceTest.cpp)."""
import numpy as np
import pytest

import hostutil as hu
from oracle import p1_oracle as po

MESHES = ["pyramid_2el", "regular_octahedron_8el", "cube_6el", "pyramid_tilted_4el"]


def _fields(glob, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(glob.ndof), rng.standard_normal(glob.ndof)


@pytest.mark.parametrize("mesh", MESHES)
@pytest.mark.parametrize("level", [1, 2, 3])
@pytest.mark.parametrize("backwards", [False, True])
def test_cell_centric_sweep_equals_the_global_schedule(mesh, level, backwards):
    v, c = hu.read_msh(hu.MESHES / f"{mesh}.msh")
    glob = hu.GlobalSweepOracle(v, c, level)
    u, b = _fields(glob, 7 + level)
    masks = hu.dirichlet_masks(v, c)
    for relax in (1.0, 1.3):
        want = glob.sweep(u, b, relax, backwards)
        got = hu.CellCentricSweep(v, c, level).sweep(glob, glob.to_cells(u), glob.to_cells(b), masks, relax, backwards)
        for g, a in zip(glob.gidx, got):
            np.testing.assert_allclose(a, want[g], rtol=0, atol=1e-12 * np.abs(want).max())


@pytest.mark.parametrize("mesh", ["tet_1el", "pyramid_2el"])
def test_sweep_over_all_points_including_the_domain_boundary(mesh):
    """flag All (e.g. a mass-matrix smoother): boundary primitives with a single neighbour cell are swept too"""
    v, c = hu.read_msh(hu.MESHES / f"{mesh}.msh")
    level = 3
    glob = hu.GlobalSweepOracle(v, c, level, form=1)
    u, b = _fields(glob, 3)
    masks = hu.dirichlet_masks(v, c, all_points=True)
    want = glob.sweep(u, b, 1.0, False, dirichlet=False)
    got = hu.CellCentricSweep(v, c, level, form=1).sweep(glob, glob.to_cells(u), glob.to_cells(b), masks, 1.0, False)
    for g, a in zip(glob.gidx, got):
        np.testing.assert_allclose(a, want[g], rtol=0, atol=1e-12 * np.abs(want).max())


def test_gauss_seidel_is_a_fixed_point_iteration_of_the_global_system():
    """A u = b  =>  one sweep leaves u unchanged, forwards and backwards"""
    v, c = hu.read_msh(hu.MESHES / "regular_octahedron_8el.msh")
    glob = hu.GlobalSweepOracle(v, c, 2)
    rng = np.random.default_rng(0)
    u = rng.standard_normal(glob.ndof)
    b = glob.matvec(u)
    masks = hu.dirichlet_masks(v, c)
    for backwards in (False, True):
        got = hu.CellCentricSweep(v, c, 2).sweep(glob, glob.to_cells(u), glob.to_cells(b), masks, 1.0, backwards)
        for g, a in zip(glob.gidx, got):
            np.testing.assert_allclose(a, u[g], rtol=0, atol=1e-13)


def test_copies_of_shared_points_stay_bit_identical():
    v, c = hu.read_msh(hu.MESHES / "regular_octahedron_8el.msh")
    glob = hu.GlobalSweepOracle(v, c, 3)
    u, b = _fields(glob, 11)
    got = hu.CellCentricSweep(v, c, 3).sweep(glob, glob.to_cells(u), glob.to_cells(b), hu.dirichlet_masks(v, c), 1.0, False)
    ref = glob.to_global(got)
    for g, a in zip(glob.gidx, got):
        assert np.array_equal(a, ref[g])
