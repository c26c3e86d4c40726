"""Timing tree with the reference's timer names and JSON layout (walberla::WcTimingTree threaded through
PrimitiveStorage::getTimingTree(), Operator.hpp:148-166, GeometricMultigridSolver.hpp:200-300, dataexport/TimingOutput.hpp)."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
STAT_KEYS = {"total", "average", "count", "min", "max", "variance"}


def _children(node):
    return {k: v for k, v in node.items() if k not in STAT_KEYS}


def test_timing_tree_names_nesting_and_json_layout():
    import torch

    from hyteg_amd import host

    assert torch.cuda.is_available()
    st = host.Storage.from_gmsh(ROOT / "hyteg_amd" / "data" / "meshes" / "regular_octahedron_8el.msh")
    with pytest.raises(host.HytegHostError, match="not enabled"):
        st.timing_json()
    min_level, max_level, cycles = 2, 4, 3
    A = host.P1ConstantOperator(st, min_level, max_level)
    A.compute_inverse_diagonal()
    u, b, r = (host.P1Function(st, n, min_level, max_level) for n in ("u", "b", "r"))
    u.interpolate(1.0, max_level, host.Inner)
    gmg = host.Solver.gmg(st, min_level, max_level, smoother=host.JACOBI, relax=2.0 / 3.0, pre=2, post=1)
    gmg.solve(A, u, b, max_level)  # untimed: lazily built tables
    st.enable_timing(True, synchronize=True)
    for _ in range(cycles):
        gmg.solve(A, u, b, max_level)
    A.apply(u, r, max_level, host.Inner)
    r.dot(r, max_level, host.Inner)
    tree = json.loads(st.timing_json())

    def check(node):  # every node carries walberla's six statistics, consistent with each other
        for name, child in _children(node).items():
            assert STAT_KEYS <= set(child), name
            assert child["count"] >= 1 and child["min"] <= child["average"] <= child["max"] + 1e-15
            assert abs(child["total"] - child["average"] * child["count"]) <= 1e-9 * max(child["total"], 1e-30)
            assert child["variance"] >= 0.0
            check(child)

    check(tree)
    top = _children(tree)
    gm = top["Geometric Multigrid Solver"]
    assert gm["count"] == cycles
    levels = _children(gm)
    assert set(levels) == {f"Level {l}" for l in range(min_level, max_level + 1)}
    fine = _children(levels[f"Level {max_level}"])
    assert {"Smoother", "Residual", "Restriction", "Prolongation"} <= set(fine)
    assert fine["Smoother"]["count"] == 2 * cycles  # pre and post smoothing
    assert _children(levels[f"Level {min_level}"])["Coarse Grid Solver"]["count"] == cycles
    # operator and function ranges nest inside the multigrid ranges with the reference's names
    smoother_ops = _children(_children(fine["Smoother"])["Operator P1Function to P1Function"])
    assert smoother_ops["smooth_jac"]["count"] == 3 * cycles
    residual = _children(fine["Residual"])
    assert _children(residual["Operator P1Function to P1Function"])["Apply"]["count"] == cycles
    assert "Assign" in _children(residual["P1Function"])
    # the stand-alone calls after the cycles are top-level ranges
    assert _children(top["Operator P1Function to P1Function"])["Apply"]["count"] == 1
    fn = _children(top["P1Function"])
    assert fn["Dot (local)"]["count"] == 1 and fn["Dot (reduce)"]["count"] == 1
    # synchronised ranges measure execution: a cycle takes longer than the sum of nothing, and parents cover children
    assert gm["total"] >= sum(c["total"] for c in levels.values()) * 0.999
    st.timing_reset()
    assert _children(json.loads(st.timing_json())) == {}
    st.enable_timing(False)
    for o in (gmg, u, b, r, A, st):
        o.close()
