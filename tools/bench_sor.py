#!/usr/bin/env python3
"""Timings of the macro-cell SOR / Gauss-Seidel sweep at levels 5-8, forward and backward, blocked form
(default) and, with --all, the dataflow form.  Usage: python tools/bench_sor.py [--levels 5 6 7 8] [--reps 10]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from hyteg_amd import capi, host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, nargs="+", default=[5, 6, 7, 8])
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--all", action="store_true")
    args = ap.parse_args()
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream
    st = host.Storage.from_gmsh(ROOT / "hyteg_amd/data/meshes/tet_1el.msh")
    st.set_stream(sh)
    # "default": what the entry point picks by level (one-workgroup LDS kernel up to level 4, blocks above)
    algs = [("default", capi.SOR_AUTO)] + ([("blocks", capi.SOR_BLOCKS), ("dataflow", capi.SOR_DATAFLOW)] if args.all else [])
    for L in args.levels:
        n = capi.cell_size(L)
        op = host.P1ConstantOperator(st, 2, L)
        w = list(op.stencils(0, L)[0])
        nbuf = max(2, min(16, int(1.5 * 256 * 2**20) // (2 * n * 8) + 1))
        U = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
        R = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
        for name, alg in algs:
            capi.set_sor_algorithm(alg)
            for bw in (False, True):
                for k in range(2):
                    capi.p1_sor_cell(U[k % nbuf].data_ptr(), R[k % nbuf].data_ptr(), L, w, 1.0, bw, sh)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for k in range(args.reps):
                    capi.p1_sor_cell(U[k % nbuf].data_ptr(), R[k % nbuf].data_ptr(), L, w, 1.0, bw, sh)
                e1.record(stream)
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / args.reps
                print(f"level {L} {name:9s} {'backward' if bw else 'forward ':8s} {us:9.1f} us per sweep  "
                      f"{capi.cell_inner_size(L) / us * 1e-3:6.2f} G DoF-updates/s", flush=True)
    capi.set_sor_algorithm(capi.SOR_AUTO)


if __name__ == "__main__":
    main()
