#!/usr/bin/env python3
"""Where a V(3,3) cycle spends its time: the timing tree (synchronised ranges) per level and phase.
Usage: python tools/vcycle_breakdown.py [--mesh tet_1el] [--min 2] [--max 7] [--smoother gs|jacobi]"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from hyteg_amd import host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="tet_1el")
    ap.add_argument("--min", type=int, default=2)
    ap.add_argument("--max", type=int, default=7)
    ap.add_argument("--smoother", default="gs")
    ap.add_argument("--cycles", type=int, default=5)
    a = ap.parse_args()
    st = host.Storage.from_gmsh(ROOT / f"hyteg_amd/data/meshes/{a.mesh}.msh")
    st.set_stream(torch.cuda.current_stream().cuda_stream)
    A = host.P1ConstantOperator(st, a.min, a.max)
    A.compute_inverse_diagonal()
    u, b = (host.P1Function(st, n, a.min, a.max) for n in ("u", "b"))
    u.interpolate(1.0, a.max, host.Inner)
    sm = host.GAUSS_SEIDEL if a.smoother == "gs" else host.JACOBI
    gmg = host.Solver.gmg(st, a.min, a.max, smoother=sm, relax=2.0 / 3.0 if a.smoother != "gs" else 1.0, pre=3, post=3)
    for _ in range(2):
        gmg.solve(A, u, b, a.max)
    st.enable_timing(True, synchronize=True)
    for _ in range(a.cycles):
        gmg.solve(A, u, b, a.max)
    tree = json.loads(st.timing_json())

    def kids(n):
        return {k: v for k, v in n.items() if isinstance(v, dict)}

    gm = kids(tree)["Geometric Multigrid Solver"]
    print(f"{a.mesh} levels {a.min}-{a.max} V(3,3) {a.smoother}: {gm['total'] / a.cycles * 1e3:.3f} ms per cycle (synchronised ranges: slower than the free-running cycle)")
    for lname, lv in sorted(kids(gm).items()):
        parts = ", ".join(f"{k} {v['total'] / a.cycles * 1e6:8.1f} us" for k, v in kids(lv).items())
        print(f"  {lname}: {lv['total'] / a.cycles * 1e6:9.1f} us  ({parts})")


if __name__ == "__main__":
    main()
