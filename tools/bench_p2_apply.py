#!/usr/bin/env python3
"""P2 elementwise / constant Laplace apply on one macro-cell: the inner DoFs, the boundary DoFs and both.
Usage: python tools/bench_p2_apply.py [--levels 5 6 7] [--reps 20]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from hyteg_amd import capi, host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, nargs="+", default=[5, 6, 7])
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream
    st = host.Storage.from_gmsh(ROOT / "hyteg_amd/data/meshes/tet_1el.msh")
    st.set_stream(sh)
    for L in args.levels:
        nv, ne = capi.cell_size(L), capi.p2_edge_array_size(L)
        op = host.P2ElementwiseLaplaceOperator(st, L, L)
        em = torch.from_numpy(capi.p2_build_operator_table(op.element_matrices(L))).to("cuda")
        nb = max(2, min(12, int(1.5 * 256 * 2**20) // (2 * (nv + ne) * 8) + 1))
        SV = [torch.rand(nv, dtype=torch.float64, device="cuda") for _ in range(nb)]
        SE = [torch.rand(ne, dtype=torch.float64, device="cuda") for _ in range(nb)]
        DV = [torch.zeros(nv, dtype=torch.float64, device="cuda") for _ in range(nb)]
        DE = [torch.zeros(ne, dtype=torch.float64, device="cuda") for _ in range(nb)]
        for name, mask in (("inner DoFs", 0x4000), ("boundary DoFs", 0x3FFF), ("all DoFs", 0x7FFF)):
            def fn(k):
                capi.p2_elementwise_apply_cell(DV[k % nb].data_ptr(), DE[k % nb].data_ptr(), SV[k % nb].data_ptr(), SE[k % nb].data_ptr(),
                                               L, em.data_ptr(), 1.0, 0, mask, sh)
            for k in range(3):
                fn(k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for k in range(args.reps):
                fn(k)
            e1.record(stream)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.reps
            print(f"level {L} P2 apply, {name:14s} {us:8.2f} us   ({nv + ne} DoFs, {16 * (nv + ne) / us * 1e-3:7.1f} GB/s algorithmic if all)", flush=True)
        # one destination kind at a time (the per-type sweeps of the Gauss-Seidel smoother) and a whole sweep through the host layer
        for kinds, name in ((0x01, "vertex DoFs only"), (0x02, "X edges only"), (0x80, "XYZ edges only")):
            def fk(k):
                capi.p2_elementwise_apply_cell(DV[k % nb].data_ptr(), DE[k % nb].data_ptr(), SV[k % nb].data_ptr(), SE[k % nb].data_ptr(),
                                               L, em.data_ptr(), 1.0, 0, 0x4000, sh, kinds)
            for k in range(3):
                fk(k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for k in range(args.reps):
                fk(k)
            e1.record(stream)
            torch.cuda.synchronize()
            print(f"level {L} P2 apply, inner, {name:17s} {e0.elapsed_time(e1) * 1e3 / args.reps:8.2f} us", flush=True)
        op.compute_inverse_diagonal()
        x, b = host.P2Function(st, "x", L, L), host.P2Function(st, "b", L, L)
        x.interpolate(1.0, L, host.Inner)
        b.interpolate(0.0, L)
        for k in range(2):
            op.smooth_sor(x, b, 1.0, L, host.Inner)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(5):
            op.smooth_sor(x, b, 1.0, L, host.Inner)
        e1.record(stream)
        torch.cuda.synchronize()
        print(f"level {L} P2 Gauss-Seidel sweep (host layer, one macro-cell) {e0.elapsed_time(e1) * 1e3 / 5:9.1f} us", flush=True)
        x.close(), b.close()


if __name__ == "__main__":
    main()
