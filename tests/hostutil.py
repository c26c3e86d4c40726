"""Test helpers for the multi-cell host layer: a numpy restatement of the cell-centric multi-cell algorithms built
from the single-cell CPU oracle (oracle/p1_oracle) and the exchange plans exported by the host library."""
from pathlib import Path

import numpy as np

from hyteg_amd import host
from oracle import p1_oracle as po

MESHES = Path(__file__).resolve().parent.parent / "hyteg_amd" / "data" / "meshes"


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


# ---------------------------------------------------------------------------------------------------------------
# Multi-cell SOR / Gauss-Seidel: (1) a global-matrix restatement of the reference's schedule, independent of the
# cell-centric decomposition, (2) the cell-centric tables (total weights, sweep orientations) in numpy.
# ---------------------------------------------------------------------------------------------------------------
CELL_EDGE_VERTS = [(0, 1), (0, 2), (1, 2), (0, 3), (1, 3), (2, 3)]
CELL_FACE_VERTS = [(0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)]
OFFS = np.array([[0, 0, -1], [1, 0, -1], [-1, 1, -1], [0, 1, -1], [0, -1, 0], [1, -1, 0], [-1, 0, 0], [0, 0, 0],
                 [1, 0, 0], [-1, 1, 0], [0, 1, 0], [0, -1, 1], [1, -1, 1], [-1, 0, 1], [0, 0, 1]])
FACE_DIRS = [(-1, 0), (1, 0), (0, -1), (0, 1), (1, -1), (-1, 1)]
UNIT = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])


def read_msh(path):
    """Gmsh 2.2 / 4.1 ASCII: (vertices (n,3), tetrahedra (m,4) as 0-based indices in node order)"""
    lines = [ln.strip() for ln in Path(path).read_text().split("\n")]
    v41 = lines[lines.index("$MeshFormat") + 1].startswith("4")
    ids, xyz, cells = [], [], []
    i = lines.index("$Nodes") + 1
    if v41:
        nblocks = int(lines[i].split()[0])
        i += 1
        for _ in range(nblocks):
            cnt = int(lines[i].split()[3])
            ids += [int(t) for t in lines[i + 1:i + 1 + cnt]]
            xyz += [[float(x) for x in ln.split()[:3]] for ln in lines[i + 1 + cnt:i + 1 + 2 * cnt]]
            i += 1 + 2 * cnt
    else:
        n = int(lines[i])
        for ln in lines[i + 1:i + 1 + n]:
            t = ln.split()
            ids.append(int(t[0]))
            xyz.append([float(x) for x in t[1:4]])
    pos = {nid: k for k, nid in enumerate(ids)}
    i = lines.index("$Elements") + 1
    if v41:
        nblocks = int(lines[i].split()[0])
        i += 1
        for _ in range(nblocks):
            _, _, etype, cnt = (int(t) for t in lines[i].split())
            if etype == 4:
                cells += [[pos[int(t)] for t in ln.split()[1:5]] for ln in lines[i + 1:i + 1 + cnt]]
            i += 1 + cnt
    else:
        m = int(lines[i])
        for ln in lines[i + 1:i + 1 + m]:
            t = [int(x) for x in ln.split()]
            if t[1] == 4:
                cells.append([pos[x] for x in t[3 + t[2]:3 + t[2] + 4]])
    return np.array(xyz, dtype=np.float64), np.array(cells, dtype=np.int64)


def _offset_index(d):
    for k in range(15):
        if tuple(OFFS[k]) == tuple(int(v) for v in d):
            return k
    raise KeyError(d)


class GlobalSweepOracle:
    """The reference's smooth_sor on a whole mesh (P1Operator.hpp:348-418), written on a global matrix:
    macro-vertices, then macro-edges (ascending along the edge), macro-faces (rows ascending, x ascending), macro-cells
    (array order).  A neighbour value is the current one if the neighbour lies on the primitive being swept or on its
    boundary, else what the last communication delivered: the pre-sweep value (forward), the value at the start of the
    primitive class (backwards, where every class is preceded by one communication)."""

    def __init__(self, vertices, cells, level, form=0):
        self.vertices, self.mesh_cells, self.level = np.asarray(vertices, float), np.asarray(cells, int), level
        N = (1 << level) + 1
        n = N - 1
        ijk = po.cell_coords(level).astype(int)
        pos = {tuple(p): k for k, p in enumerate(ijk)}
        slots = po.slot_of_points(level)
        self.index, self.support, self.order = {}, [], []
        self.gidx = []
        face_count = {}
        for cv in self.mesh_cells:
            for f in CELL_FACE_VERTS:
                key = tuple(sorted(int(cv[a]) for a in f))
                face_count[key] = face_count.get(key, 0) + 1
        bverts = set(v for f, c in face_count.items() if c == 1 for v in f)
        bedges = set(tuple(sorted((f[a], f[b]))) for f, c in face_count.items() if c == 1 for a in range(3) for b in range(a + 1, 3))
        bfaces = set(f for f, c in face_count.items() if c == 1)
        self.boundary = []
        for c, cv in enumerate(self.mesh_cells):
            g = np.empty(len(ijk), dtype=np.int64)
            for k, (x, y, z) in enumerate(ijk):
                bary = (n - x - y - z, x, y, z)
                key = tuple(sorted((int(cv[a]), int(bary[a])) for a in range(4) if bary[a] > 0))
                if key not in self.index:
                    self.index[key] = len(self.support)
                    sup = tuple(v for v, _ in key)
                    wt = dict(key)
                    self.support.append(sup)
                    if len(sup) == 1:
                        self.order.append((0,))
                        self.boundary.append(sup[0] in bverts)
                    elif len(sup) == 2:
                        self.order.append((wt[sup[1]],))
                        self.boundary.append(sup in bedges)
                    elif len(sup) == 3:
                        self.order.append((wt[sup[2]], wt[sup[1]]))
                        self.boundary.append(sup in bfaces)
                    else:
                        self.order.append((c, k))
                        self.support[-1] = ("cell", c) + sup
                        self.boundary.append(False)
                g[k] = self.index[key]
            self.gidx.append(g)
        self.ndof = len(self.support)
        rows = [dict() for _ in range(self.ndof)]
        for c, cv in enumerate(self.mesh_cells):
            co = self.vertices[cv].reshape(12)
            w = po.assemble_cell_stencil(co, level, form)
            ws = po.assemble_cell_slot_stencils(co, level, form).reshape(14, 15)
            g = self.gidx[c]
            for k, p in enumerate(ijk):
                wk = w if slots[k] == 14 else ws[slots[k]]
                for d in range(15):
                    q = tuple(p + OFFS[d])
                    if q in pos and wk[d] != 0.0:
                        rows[g[k]][g[pos[q]]] = rows[g[k]].get(g[pos[q]], 0.0) + wk[d]
        self.rows = rows

    def closure_contains(self, p, j):
        sp, sj = self.support[p], self.support[j]
        if sp[0] == "cell":
            return True
        if sj[0] == "cell":
            return False
        return set(sj) <= set(sp)

    def dim(self, p):
        s = self.support[p]
        return 4 if s[0] == "cell" else len(s)

    def to_global(self, arrays):
        u = np.zeros(self.ndof)
        for g, a in zip(self.gidx, arrays):
            u[g] = a
        return u

    def to_cells(self, u):
        return [u[g].copy() for g in self.gidx]

    def matvec(self, u):
        return np.array([sum(w * u[j] for j, w in r.items()) for r in self.rows])

    def sweep(self, u, b, relax=1.0, backwards=False, dirichlet=True):
        u = u.copy()
        snap = u.copy()
        dims = [4, 3, 2, 1] if backwards else [1, 2, 3, 4]
        for dm in dims:
            if backwards:
                snap = u.copy()
            prims = {}
            for p in range(self.ndof):
                if self.dim(p) == dm and not (dirichlet and self.boundary[p]):
                    prims.setdefault(self.support[p], []).append(p)
            for sup, pts in prims.items():
                pts.sort(key=lambda p: self.order[p], reverse=backwards)
                for p in pts:
                    tmp = b[p]
                    for j, w in self.rows[p].items():
                        if j != p:
                            tmp -= w * (u[j] if self.closure_contains(p, j) else snap[j])
                    u[p] = (1.0 - relax) * u[p] + relax * tmp / self.rows[p][p]
        return u


def sor_tables(vertices, cells, level, form=0):
    """per cell: dict(rest_slots (14,15), edge_verts, edge_w, face_verts, face_w, vertex_w) of hyteg_hip_p1_sor_shell_cell"""
    vertices, cells = np.asarray(vertices, float), np.asarray(cells, int)
    slot_w = [po.assemble_cell_slot_stencils(vertices[cv].reshape(12), level, form).reshape(14, 15) for cv in cells]
    vert_c, edge_t, face_t = {}, {}, {}
    per_cell = []
    for c, cv in enumerate(cells):
        info = dict(edge_verts=[], face_verts=[], edge_dirs=[], face_dirs=[])
        for k in range(4):
            vert_c[int(cv[k])] = vert_c.get(int(cv[k]), 0.0) + slot_w[c][10 + k][7]
        for e, (a, b) in enumerate(CELL_EDGE_VERTS):
            lo, hi = (a, b) if cv[a] < cv[b] else (b, a)
            d = UNIT[hi] - UNIT[lo]
            kp, km = _offset_index(d), _offset_index(-d)
            key = (int(cv[lo]), int(cv[hi]))
            t = edge_t.setdefault(key, np.zeros(3))
            t += [slot_w[c][e][7], slot_w[c][e][km], slot_w[c][e][kp]]
            info["edge_verts"].append((lo, hi))
            info["edge_dirs"].append((km, kp))
        for f, loc in enumerate(CELL_FACE_VERTS):
            a, b, cc = sorted(loc, key=lambda v: cv[v])
            d1, d2 = UNIT[b] - UNIT[a], UNIT[cc] - UNIT[a]
            ks = [_offset_index(i * d1 + j * d2) for i, j in FACE_DIRS]
            key = (int(cv[a]), int(cv[b]), int(cv[cc]))
            t = face_t.setdefault(key, np.zeros(7))
            t += [slot_w[c][6 + f][7]] + [slot_w[c][6 + f][k] for k in ks]
            info["face_verts"].append((a, b, cc))
            info["face_dirs"].append(ks)
        per_cell.append(info)
    out = []
    for c, cv in enumerate(cells):
        info = per_cell[c]
        rest = slot_w[c].copy()
        rest[:, 7] = 0.0
        for e in range(6):
            rest[e, list(info["edge_dirs"][e])] = 0.0
        for f in range(4):
            rest[6 + f, info["face_dirs"][f]] = 0.0
        out.append(dict(
            rest_slots=rest,
            edge_verts=info["edge_verts"],
            edge_w=[edge_t[(int(cv[lo]), int(cv[hi]))] for lo, hi in info["edge_verts"]],
            face_verts=info["face_verts"],
            face_w=[face_t[tuple(int(cv[v]) for v in fv)] for fv in info["face_verts"]],
            vertex_w=[vert_c[int(cv[k])] for k in range(4)]))
    return out


def dirichlet_masks(vertices, cells, all_points=False):
    """per cell: 15-bit mask of the points a sweep with flag Inner touches when the whole boundary is Dirichlet"""
    cells = np.asarray(cells, int)
    count = {}
    for cv in cells:
        for f in CELL_FACE_VERTS:
            key = tuple(sorted(int(cv[a]) for a in f))
            count[key] = count.get(key, 0) + 1
    bfaces = set(f for f, c in count.items() if c == 1)
    bverts = set(v for f in bfaces for v in f)
    bedges = set(tuple(sorted((f[a], f[b]))) for f in bfaces for a in range(3) for b in range(a + 1, 3))
    masks = []
    for cv in cells:
        m = po.MASK_INNER
        for e, (a, b) in enumerate(CELL_EDGE_VERTS):
            if all_points or tuple(sorted((int(cv[a]), int(cv[b])))) not in bedges:
                m |= 1 << e
        for f, loc in enumerate(CELL_FACE_VERTS):
            if all_points or tuple(sorted(int(cv[a]) for a in loc)) not in bfaces:
                m |= 1 << (6 + f)
        for k in range(4):
            if all_points or int(cv[k]) not in bverts:
                m |= 1 << (10 + k)
        masks.append(m)
    return masks


class CellCentricSweep:
    """smooth_sor the way the host layer composes it from per-cell operations; `ops` supplies the per-cell kernels
    (the CPU oracle's or the HIP library's), so the same driver checks the decomposition on the CPU and the kernels on the GPU."""

    def __init__(self, vertices, cells, level, form=0):
        self.vertices, self.cells, self.level = np.asarray(vertices, float), np.asarray(cells, int), level
        self.tables = sor_tables(vertices, cells, level, form)
        self.inner = [po.assemble_cell_stencil(self.vertices[cv].reshape(12), level, form) for cv in self.cells]

    def sweep(self, glob: GlobalSweepOracle, u, b, masks, relax=1.0, backwards=False):
        """u, b: lists of cell arrays (consistent copies); returns the new list.  Copies are summed through glob's numbering."""
        shell = po.slot_of_points(self.level) < 14

        def sum_copies(rest):
            total = np.zeros(glob.ndof)
            for c in range(len(rest)):
                np.add.at(total, glob.gidx[c][shell], rest[c][shell])
            for c in range(len(rest)):
                rest[c][:] = total[glob.gidx[c]]

        return self.sweep_with(sum_copies, u, b, masks, relax, backwards)

    def sweep_with(self, sum_copies, u, b, masks, relax=1.0, backwards=False):
        """sum_copies(list of cell arrays): in-place sum over the copies of every shared point"""
        L = self.level
        u = [a.copy() for a in u]

        def rest_pass(bits):
            rest = [np.zeros_like(a) for a in u]
            for c, t in enumerate(self.tables):
                po.apply_cell_boundary(rest[c], u[c], L, t["rest_slots"].reshape(-1), masks[c] & bits, po.REPLACE)
            sum_copies(rest)
            return rest

        def shell(rest, bits):
            for c, t in enumerate(self.tables):
                po.sor_shell_cell(u[c], b[c], rest[c], L, t["edge_verts"], t["edge_w"], t["face_verts"], t["face_w"], t["vertex_w"],
                                  relax, masks[c] & bits, backwards)

        def cells():
            if L >= 2:
                for c in range(len(u)):
                    if masks[c] & po.MASK_INNER:
                        po.sor_cell(u[c], b[c], L, self.inner[c], relax, backwards)

        if not backwards:
            shell(rest_pass(po.MASK_SHELL), po.MASK_SHELL)
            cells()
        else:
            cells()
            for bits in (0xF << 6, 0x3F, 0xF << 10):
                shell(rest_pass(bits), bits)
        return u
