#!/usr/bin/env python3
"""Brick-shape sweep of the z-march kernels (hyteg_hip_set_apply_shape): every compiled shape x mode x level, rotating
buffers larger than the Infinity Cache, HIP events around `reps` launches, best of `rounds` rounds.  Evidence for the
per-level defaults in hyteg_amd/csrc/p1_apply.hip.  Usage: python tools/apply_shape_sweep.py [--levels 6 7 8 9]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from hyteg_amd import capi  # noqa: E402

SHAPES = [(2, 8, 1), (4, 8, 2), (4, 8, 1), (4, 4, 2), (4, 4, 1), (2, 4, 1), (8, 4, 2)]  # HYTEG_ZM_SHAPES of p1_apply.hip (2x8x2 was in the first sweep)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, nargs="+", default=[6, 7, 8, 9])
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream
    w = [0.1 * (k + 1) for k in range(15)]
    w[7] = -3.0
    for L in args.levels:
        n, inner = capi.cell_size(L), capi.cell_inner_size(L)
        capi.prepare_level(L)
        # the SOURCE arrays alone (also the float ones) exceed the 256 MiB Infinity Cache 2.2 times: the nontemporal stores of the
        # destination do not stay in that cache, so a ring that merely exceeds it in total is read from it (round 3)
        nbuf = max(3, -(-int(2.2 * 256 * 2**20) // (n * 4)))
        nbuf = min(nbuf, 64)  # small levels: cache-resident either way
        A = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
        B = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
        Cc = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
        Af, Bf, Cf = [t.float() for t in A], [t.float() for t in B], [t.float() for t in Cc]
        p = lambda t, k: t[k % nbuf].data_ptr()  # noqa: E731
        modes = {
            "apply Replace": lambda k: capi.p1_apply_cell(p(B, k), p(A, k), L, w, 0, sh),
            "apply Add": lambda k: capi.p1_apply_cell(p(B, k), p(A, k), L, w, 1, sh),
            "Jacobi scalar invdiag": lambda k: capi.p1_jacobi_cell(p(B, k), p(Cc, k), p(A, k), L, w, 0.66, None, sh),
            "Jacobi invdiag function": lambda k: capi.p1_jacobi_cell(p(B, k), p(Cc, k), p(A, k), L, w, 0.66, p(Cc, k + 1), sh),
            "residual": lambda k: capi.p1_residual_cell(p(B, k), p(Cc, k), p(A, k), L, w, sh),
            "apply Replace f32": lambda k: capi.p1_apply_cell_f32(p(Bf, k), p(Af, k), L, w, 0, sh),
            "Jacobi scalar invdiag f32": lambda k: capi.p1_jacobi_cell_f32(p(Bf, k), p(Cf, k), p(Af, k), L, w, 0.66, None, sh),
        }
        reps = args.reps if L >= 8 else 2 * args.reps
        if L >= 9:
            reps = max(20, args.reps // 6)
        for name, fn in modes.items():
            res = {}
            for rnd in range(args.rounds):
                for s in SHAPES:
                    capi.set_apply_shape(*s)
                    for k in range(5):
                        fn(k)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for k in range(reps):
                        fn(k)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / reps
                    res[s] = min(res.get(s, 1e9), us)
            capi.set_apply_shape()
            best = min(res, key=res.get)
            print(f"level {L} {name:26s} " + "  ".join(f"{s[0]}x{s[1]}x{s[2]} {res[s]:7.2f}" for s in SHAPES)
                  + f"   best {best[0]}x{best[1]}x{best[2]}", flush=True)
        del A, B, Cc, Af, Bf, Cf
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
