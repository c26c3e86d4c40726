"""Test helpers for the multi-cell host layer: a numpy restatement of the cell-centric multi-cell algorithms built
from the single-cell CPU oracle (oracle/p1_oracle) and the exchange plans exported by the host library."""
from pathlib import Path

import numpy as np

from hyteg_amd import host
from oracle import p1_oracle as po

MESHES = Path(__file__).resolve().parent / "golden" / "meshes"


def cell_points(coords4, level):
    """(size,3) physical coordinates of the micro-vertices of a cell in array order (VertexDoFMacroCell.hpp:70-77)"""
    cc = np.asarray(coords4, dtype=np.float64).reshape(4, 3)
    ijk = po.cell_coords(level).astype(np.float64)
    step = 1.0 / float(1 << level)
    xs, ys, zs = (cc[1] - cc[0]) * step, (cc[2] - cc[0]) * step, (cc[3] - cc[0]) * step
    return cc[0][None, :] + ijk[:, 0:1] * xs[None, :] + ijk[:, 1:2] * ys[None, :] + ijk[:, 2:3] * zs[None, :]


def point_mask(level, mask):
    """boolean per array entry: selected by the 15-bit point mask"""
    return ((mask >> po.slot_of_points(level)) & 1).astype(bool)


class MultiCellOracle:
    """single-rank numpy model: arrays[c] is the cell array of local cell c"""

    def __init__(self, storage: host.Storage):
        assert storage.n_ranks == 1
        self.st = storage
        self.cells = [storage.local_cell(i) for i in range(storage.n_local_cells)]  # (gid, coords, nnc)

    def interpolate(self, fn, level):
        out = []
        for gid, co, nnc in self.cells:
            p = cell_points(co, level)
            out.append(np.ascontiguousarray(fn(p[:, 0], p[:, 1], p[:, 2]), dtype=np.float64))
        self.sync(out, level, host.All)
        return out

    def _plans(self, level, flag):
        for cls, typ in ((0, host.Inner), (1, self.boundary_type)):
            if typ & flag:
                yield self.st.plan(level, cls)

    boundary_type = host.DirichletBoundary

    def sum_shared(self, arrays, level, flag):
        for p in self._plans(level, flag):
            gp, eb, eo = p["group_ptr"], p["entry_buf"], p["entry_off"]
            for g in range(p["ngroups"]):
                s = 0.0
                for e in range(gp[g], gp[g + 1]):
                    s = arrays[eb[e]][eo[e]] if e == gp[g] else s + arrays[eb[e]][eo[e]]
                for e in range(gp[g], gp[g + 1]):
                    arrays[eb[e]][eo[e]] = s

    def sync(self, arrays, level, flag):
        for p in self._plans(level, flag):
            gp, eb, eo = p["group_ptr"], p["entry_buf"], p["entry_off"]
            for g in range(p["ngroups"]):
                v = arrays[eb[gp[g]]][eo[gp[g]]]
                for e in range(gp[g], gp[g + 1]):
                    arrays[eb[e]][eo[e]] = v

    def apply(self, src, dst, level, flag, form=0):
        """Replace-mode apply; dst entries outside the flag are left as they are"""
        for i, (gid, co, nnc) in enumerate(self.cells):
            m = self.st.mask(i, flag)
            w = po.assemble_cell_stencil(co, level, form)
            ws = po.assemble_cell_slot_stencils(co, level, form)
            if (m & po.MASK_INNER) and level >= 2:
                po.apply_cell(dst[i], src[i], level, w)
            po.apply_cell_boundary(dst[i], src[i], level, ws, m & po.MASK_SHELL)
        self.sum_shared(dst, level, flag)
        return dst

    def dot(self, a, b, level, flag):
        return sum(po.dot_cell_masked(a[i], b[i], level, self.st.mask(i, flag, owned=True)) for i in range(len(self.cells)))


def upload(f: host.P1Function, arrays, level):
    for c, a in enumerate(arrays):
        f.upload_cell(c, level, a)


def download(f: host.P1Function, level):
    return [f.download_cell(c, level) for c in range(f.storage.n_local_cells)]
