"""numpy/ctypes front end of the CPU oracle (oracle/p1_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py may import this module.  The product package ``hyteg_amd``
never does (tests/test_product_isolation.py enforces it).

Each wrapper mirrors one ``ho_*`` function; the reference file:line each restates is cited in
p1_oracle.c.  Arrays are float64, C-contiguous, in HyTeG's linear tetrahedral cell layout.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_P = C.POINTER(C.c_double)

# stencil slot -> logical offset, the C-ABI order (std::map<Index> iteration order: z, y, x)
STENCIL_OFFSETS = (
    (0, 0, -1), (1, 0, -1), (-1, 1, -1), (0, 1, -1),
    (0, -1, 0), (1, -1, 0), (-1, 0, 0), (0, 0, 0), (1, 0, 0), (-1, 1, 0), (0, 1, 0),
    (0, -1, 1), (1, -1, 1), (-1, 0, 1), (0, 0, 1),
)
# HyTeG stencilDirection names of the 15 slots (src/hyteg/StencilDirections.hpp)
STENCIL_NAMES = ("BC", "BE", "BNW", "BN", "S", "SE", "W", "C", "E", "NW", "N", "TS", "TSE", "TW", "TC")
REPLACE, ADD = 0, 1


def _build(target: str) -> Path:
    so = _HERE / "_build" / target
    src = _HERE / "p1_oracle.c"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), f"_build/{target}"], check=True, capture_output=True)
    return so


def _bind(lib):
    i, d, ll = C.c_int, C.c_double, C.c_int64
    sig = {
        "ho_width": (ll, [i]),
        "ho_cell_size": (ll, [i]),
        "ho_cell_index": (ll, [i, i, i, i]),
        "ho_cell_inner_size": (ll, [i]),
        "ho_face_size_w": (ll, [ll]),
        "ho_face_index_w": (ll, [ll, ll, ll]),
        "ho_apply_cell": (None, [_P, _P, i, _P, i]),
        "ho_sor_cell": (None, [_P, _P, i, _P, d, i]),
        "ho_gs_cell": (None, [_P, _P, i, _P]),
        "ho_assign": (None, [_P, i, C.POINTER(_P), _P, i]),
        "ho_add": (None, [_P, i, C.POINTER(_P), _P, i]),
        "ho_add_scalar": (None, [_P, d, i]),
        "ho_mult_elementwise": (None, [_P, i, C.POINTER(_P), i]),
        "ho_dot": (d, [_P, _P, i]),
        "ho_set_inner": (None, [_P, d, i]),
        "ho_jacobi_cell": (None, [_P, _P, _P, _P, i, _P, d]),
        "ho_prim_slot": (i, [i, i, i, i]),
        "ho_restrict_cell": (None, [_P, _P, i, _P]),
        "ho_prolongate_prepare": (None, [_P, i, i]),
        "ho_prolongate_cell": (None, [_P, _P, i, _P]),
        "ho_apply_cell_f32": (None, [C.c_void_p, C.c_void_p, C.c_int, _P, C.c_int]),
        "ho_jacobi_cell_f32": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _P, C.c_double]),
        "ho_p1_tet_diffusion": (None, [_P, _P]),
        "ho_p1_tet_mass": (None, [_P, _P]),
        "ho_p1_tet_div": (None, [_P, _P, C.c_int]),
        "ho_p1_tet_divt": (None, [_P, _P, C.c_int]),
        "ho_p1_tet_pspg": (None, [_P, _P]),
        "ho_assemble_cell_stencil": (None, [_P, _P, i, i]),
        "ho_coordinate_from_index": (None, [_P, _P, i, i, i, i]),
        "ho_assemble_cell_slot_stencils": (None, [_P, _P, i, i]),
        "ho_apply_cell_boundary": (None, [_P, _P, i, _P, C.c_uint, i]),
        "ho_vector_cell_masked": (None, [i, _P, i, C.POINTER(_P), _P, i, C.c_uint]),
        "ho_dot_cell_masked": (d, [_P, _P, i, C.c_uint]),
        "ho_copy_face_to_cell": (None, [_P, _P, i, i, i, i]),
        "ho_copy_cell_to_face": (None, [_P, _P, i, i, i, i, i]),
        "ho_apply_face3d": (None, [_P, _P, i, i, C.POINTER(C.c_int), _P, i]),
        "ho_sor_face3d": (None, [_P, _P, i, i, C.POINTER(C.c_int), _P, C.c_double, i]),
        "ho_edge_array_size": (ll, [i]),
        "ho_edge_index": (ll, [i, ll, ll, ll, i]),
        "ho_p2_micro_cell_dofs": (None, [i, i, ll, ll, ll, C.POINTER(ll)]),
        "ho_p2_tet_diffusion": (None, [_P, _P]),
        "ho_p2_cell_element_matrices": (None, [_P, _P, i]),
        "ho_edge_dof_class": (i, [i, ll, ll, ll, i]),
        "ho_p2_elementwise_apply_cell": (None, [_P, _P, _P, _P, i, _P, d, i, C.c_uint]),
        "ho_sor_shell_cell": (None, [_P, _P, _P, i, C.POINTER(C.c_int), _P, C.POINTER(C.c_int), _P, _P, d, C.c_uint, i]),
        "ho_p1_elementwise_apply_macro_3d": (None, [_P, _P, _P, ll]),
        "ho_p1_elementwise_diagonal_macro_3d": (None, [_P, _P, ll]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    return lib


_lib = None
_lib_fast = None


def lib(fast: bool = False):
    """The strict (-O2, no contraction) oracle, or the -O3 -march=native build used as CPU baseline."""
    global _lib, _lib_fast
    if fast:
        if _lib_fast is None:
            _lib_fast = _bind(C.CDLL(str(_build_fast())))
        return _lib_fast
    if _lib is None:
        _lib = _bind(C.CDLL(str(_build("libp1_oracle.so"))))
    return _lib


def _build_fast() -> Path:
    """-march=native must be compiled on the machine it runs on (the GPU box's host differs from
    the dev container), so the fast build goes to a per-host temp dir, never into the repo."""
    import hashlib
    import platform
    import tempfile

    src = _HERE / "p1_oracle.c"
    tag = hashlib.sha1((platform.processor() + open("/proc/cpuinfo").read().split("\n\n")[0]).encode()
                       + src.read_bytes()).hexdigest()[:12]
    out = Path(tempfile.gettempdir()) / f"libp1_oracle_fast_{tag}.so"
    if not out.exists():
        tmp = str(out) + f".{os.getpid()}"
        subprocess.run(["gcc", "-O3", "-march=native", "-fPIC", "-fvisibility=hidden", "-shared", "-o", tmp,
                        str(src), "-lm"], check=True)
        os.replace(tmp, out)
    return out


def _p(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_P)


def _pp(arrs):
    arr = (_P * len(arrs))(*[_p(a) for a in arrs])
    return arr


def _w(w):
    w = np.ascontiguousarray(w, dtype=np.float64)
    assert w.shape == (15,)
    return w


# ---- layout ----------------------------------------------------------------------------------
def width(level): return int(lib().ho_width(level))
def cell_size(level): return int(lib().ho_cell_size(level))
def cell_inner_size(level): return int(lib().ho_cell_inner_size(level))
def cell_index(level, x, y, z): return int(lib().ho_cell_index(level, x, y, z))
def face_size_w(w): return int(lib().ho_face_size_w(w))
def face_index_w(w, x, y): return int(lib().ho_face_index_w(w, x, y))
def prim_slot(level, x, y, z): return int(lib().ho_prim_slot(level, x, y, z))


def cell_coords(level):
    """(size,3) int array of logical (x,y,z) per array slot, in memory order."""
    n = width(level)
    out = np.empty((cell_size(level), 3), dtype=np.int64)
    k = 0
    for z in range(n):
        for y in range(n - z):
            m = n - z - y
            out[k:k + m, 0] = np.arange(m)
            out[k:k + m, 1] = y
            out[k:k + m, 2] = z
            k += m
    return out


def inner_mask(level):
    c = cell_coords(level)
    n = width(level)
    return (c[:, 0] >= 1) & (c[:, 1] >= 1) & (c[:, 2] >= 1) & (c.sum(axis=1) <= n - 2)


# ---- kernels ---------------------------------------------------------------------------------
def apply_cell(dst, src, level, w, update=REPLACE, fast=False):
    lib(fast).ho_apply_cell(_p(dst), _p(src), level, _p(_w(w)), update)
    return dst


def apply_cell_f32(dst, src, level, w, update=REPLACE):
    """float32 arrays (numpy float32, contiguous)"""
    assert dst.dtype == np.float32 and src.dtype == np.float32
    lib().ho_apply_cell_f32(dst.ctypes.data, src.ctypes.data, level, _p(_w(w)), update)
    return dst


def jacobi_cell_f32(dst, rhs, src, level, w, relax, invdiag=None):
    assert dst.dtype == rhs.dtype == src.dtype == np.float32
    lib().ho_jacobi_cell_f32(dst.ctypes.data, rhs.ctypes.data, src.ctypes.data, None if invdiag is None else invdiag.ctypes.data, level,
                             _p(_w(w)), float(relax))
    return dst


def sor_cell(u, rhs, level, w, relax, backwards=False, fast=False):
    lib(fast).ho_sor_cell(_p(u), _p(rhs), level, _p(_w(w)), float(relax), int(backwards))
    return u


def gs_cell(u, rhs, level, w, fast=False):
    lib(fast).ho_gs_cell(_p(u), _p(rhs), level, _p(_w(w)))
    return u


def assign(dst, scalars, srcs, level):
    s = np.ascontiguousarray(scalars, dtype=np.float64)
    lib().ho_assign(_p(dst), len(srcs), _pp(srcs), _p(s), level)
    return dst


def add(dst, scalars, srcs, level):
    s = np.ascontiguousarray(scalars, dtype=np.float64)
    lib().ho_add(_p(dst), len(srcs), _pp(srcs), _p(s), level)
    return dst


def add_scalar(dst, scalar, level):
    lib().ho_add_scalar(_p(dst), float(scalar), level)
    return dst


def mult_elementwise(dst, srcs, level):
    lib().ho_mult_elementwise(_p(dst), len(srcs), _pp(srcs), level)
    return dst


def dot(a, b, level): return float(lib().ho_dot(_p(a), _p(b), level))


def set_inner(dst, value, level):
    lib().ho_set_inner(_p(dst), float(value), level)
    return dst


def jacobi_cell(dst, rhs, src, level, w, relax, invdiag=None, fast=False):
    lib(fast).ho_jacobi_cell(_p(dst), _p(rhs), _p(src), _p(invdiag) if invdiag is not None else None, level,
                             _p(_w(w)), float(relax))
    return dst


def _nnc(nnc):
    n = np.ascontiguousarray(nnc, dtype=np.float64)
    assert n.shape == (14,)  # edge0..5, face0..3, vertex0..3
    return n


def restrict_cell(coarse, fine, coarse_level, nnc):
    lib().ho_restrict_cell(_p(coarse), _p(fine), coarse_level, _p(_nnc(nnc)))
    return coarse


def prolongate_prepare(fine, fine_level, update):
    lib().ho_prolongate_prepare(_p(fine), fine_level, update)
    return fine


def prolongate_cell(coarse, fine, coarse_level, nnc):
    lib().ho_prolongate_cell(_p(coarse), _p(fine), coarse_level, _p(_nnc(nnc)))
    return fine


# ---- stencil assembly -------------------------------------------------------------------------
def p1_tet_diffusion(coords):
    c = np.ascontiguousarray(coords, dtype=np.float64).reshape(12)
    A = np.empty(16)
    lib().ho_p1_tet_diffusion(_p(A), _p(c))
    return A.reshape(4, 4)


def p1_tet_mass(coords):
    c = np.ascontiguousarray(coords, dtype=np.float64).reshape(12)
    A = np.empty(16)
    lib().ho_p1_tet_mass(_p(A), _p(c))
    return A.reshape(4, 4)


FORM_LAPLACE, FORM_MASS, FORM_DIV_X, FORM_DIV_Y, FORM_DIV_Z, FORM_DIVT_X, FORM_DIVT_Y, FORM_DIVT_Z, FORM_PSPG = range(9)


def p1_tet_div(coords, k):
    c = np.ascontiguousarray(coords, dtype=np.float64).reshape(12)
    A = np.empty(16)
    lib().ho_p1_tet_div(_p(A), _p(c), k)
    return A.reshape(4, 4)


def p1_tet_divt(coords, k):
    c = np.ascontiguousarray(coords, dtype=np.float64).reshape(12)
    A = np.empty(16)
    lib().ho_p1_tet_divt(_p(A), _p(c), k)
    return A.reshape(4, 4)


def p1_tet_pspg(coords):
    c = np.ascontiguousarray(coords, dtype=np.float64).reshape(12)
    A = np.empty(16)
    lib().ho_p1_tet_pspg(_p(A), _p(c))
    return A.reshape(4, 4)


def element_matrix(coords, form):
    """A[test i][trial j] of form id `form` (FORM_*)"""
    if form == FORM_LAPLACE:
        return p1_tet_diffusion(coords)
    if form == FORM_MASS:
        return p1_tet_mass(coords)
    if FORM_DIV_X <= form <= FORM_DIV_Z:
        return p1_tet_div(coords, form - FORM_DIV_X)
    if FORM_DIVT_X <= form <= FORM_DIVT_Z:
        return p1_tet_divt(coords, form - FORM_DIVT_X)
    return p1_tet_pspg(coords)


def assemble_cell_stencil(cell_vertex_coords, level, form=0):
    c = np.ascontiguousarray(cell_vertex_coords, dtype=np.float64).reshape(12)
    w = np.empty(15)
    lib().ho_assemble_cell_stencil(_p(w), _p(c), level, form)
    return w


MASK_INNER, MASK_SHELL, MASK_ALL = 1 << 14, 0x3FFF, 0x7FFF


def assemble_cell_slot_stencils(cell_vertex_coords, level, form=0):
    """(14,15) partial stencils of this cell for points on its edges 0-5, faces 0-3, vertices 0-3."""
    c = np.ascontiguousarray(cell_vertex_coords, dtype=np.float64).reshape(12)
    w = np.empty(14 * 15)
    lib().ho_assemble_cell_slot_stencils(_p(w), _p(c), level, form)
    return w.reshape(14, 15)


def apply_cell_boundary(dst, src, level, w_slots, mask=MASK_SHELL, update=REPLACE):
    ws = np.ascontiguousarray(w_slots, dtype=np.float64).reshape(14 * 15)
    lib().ho_apply_cell_boundary(_p(dst), _p(src), level, _p(ws), mask, update)
    return dst


def vector_cell_masked(op, dst, scalars, srcs, level, mask):
    s = np.ascontiguousarray(scalars if scalars is not None else [1.0] * max(1, len(srcs)), dtype=np.float64)
    lib().ho_vector_cell_masked(op, _p(dst), len(srcs), _pp(srcs) if srcs else None, _p(s), level, mask)
    return dst


def dot_cell_masked(a, b, level, mask): return float(lib().ho_dot_cell_masked(_p(a), _p(b), level, mask))


def face_array_size(level, ncells):
    n = width(level)
    return face_size_w(n) + ncells * face_size_w(n - 1)


def copy_face_to_cell(cell, face, level, v):
    lib().ho_copy_face_to_cell(_p(cell), _p(face), level, *map(int, v))
    return cell


def copy_cell_to_face(face, cell, level, v, neighbor):
    lib().ho_copy_cell_to_face(_p(face), _p(cell), level, *map(int, v), int(neighbor))
    return face


def sor_face3d(dst, rhs, level, vmaps, ws, relax, backwards=False):
    """P1Operator::smooth_sor_face3D on one macro-face in HyTeG's face layout, in place (P1Operator.hpp:1424-1503)."""
    vm = np.ascontiguousarray(np.array(vmaps, dtype=np.int32).reshape(-1))
    w = np.ascontiguousarray(np.array(ws, dtype=np.float64).reshape(-1))
    lib().ho_sor_face3d(_p(dst), _p(rhs), level, len(vmaps), vm.ctypes.data_as(C.POINTER(C.c_int)), _p(w), float(relax), int(bool(backwards)))


def apply_face3d(dst, src, level, vmaps, ws, update=REPLACE):
    vm = np.ascontiguousarray(vmaps, dtype=np.int32).reshape(-1, 3)
    w = np.ascontiguousarray(ws, dtype=np.float64).reshape(-1, 15)
    assert len(vm) == len(w)
    lib().ho_apply_face3d(_p(dst), _p(src), level, len(vm), vm.ctypes.data_as(C.POINTER(C.c_int)), _p(w.reshape(-1)), update)
    return dst


# ---- P2 (vertex + edge DoFs) on one macro-cell ----
EDGE_ORIENTATIONS = ("X", "Y", "Z", "XY", "XZ", "YZ", "XYZ")


def edge_array_size(level): return int(lib().ho_edge_array_size(level))
def edge_index(level, x, y, z, o): return int(lib().ho_edge_index(level, x, y, z, o))
def edge_dof_class(level, x, y, z, o): return int(lib().ho_edge_dof_class(level, x, y, z, o))


@functools.lru_cache(maxsize=None)
def edge_coords(level):
    """(edge array size, 4) int array: x, y, z, orientation of every edge DoF in array order (cached: treat as read-only)"""
    n = 1 << level
    out = []
    for o in range(7):
        w = n - 1 if o == 6 else n
        for z in range(w):
            for y in range(w - z):
                m = w - z - y
                if m > 0:
                    blk = np.empty((m, 4), dtype=np.int64)
                    blk[:, 0], blk[:, 1], blk[:, 2], blk[:, 3] = np.arange(m), y, z, o
                    out.append(blk)
    return np.concatenate(out).reshape(-1, 4) if out else np.zeros((0, 4), dtype=np.int64)


@functools.lru_cache(maxsize=None)
def edge_classes(level):
    """point class (0..13: the macro-primitive slot, 14: inside the cell) of every edge DoF in array order (cached)"""
    return np.array([edge_dof_class(level, int(x), int(y), int(z), int(o)) for x, y, z, o in edge_coords(level)], dtype=np.int64)


def p2_micro_cell_dofs(level, cell_type, mx, my, mz):
    out = (C.c_int64 * 10)()
    lib().ho_p2_micro_cell_dofs(level, cell_type, mx, my, mz, out)
    return list(out)


def p2_tet_diffusion(coords):
    c = np.ascontiguousarray(coords, dtype=np.float64).reshape(12)
    A = np.empty(100)
    lib().ho_p2_tet_diffusion(_p(A), _p(c))
    return A.reshape(10, 10)


def p2_cell_element_matrices(cell_vertex_coords, level):
    c = np.ascontiguousarray(cell_vertex_coords, dtype=np.float64).reshape(12)
    A = np.empty(600)
    lib().ho_p2_cell_element_matrices(_p(A), _p(c), level)
    return A.reshape(6, 10, 10)


def p2_elementwise_apply_cell(dst_v, dst_e, src_v, src_e, level, elmat, alpha=1.0, update=REPLACE, mask=0x7FFF):
    em = np.ascontiguousarray(elmat, dtype=np.float64).reshape(600)
    lib().ho_p2_elementwise_apply_cell(_p(dst_v), _p(dst_e), _p(src_v), _p(src_e), level, _p(em), float(alpha), update, mask)
    return dst_v, dst_e


def p1_elementwise_apply_macro_3d(dst, src, cell_vertex_coords, micro_edges):
    """dst += (operator of the macro-cell) src at all points: the generated elementwise kernel's micro-cell loop"""
    c = np.ascontiguousarray(cell_vertex_coords, dtype=np.float64).reshape(12)
    lib().ho_p1_elementwise_apply_macro_3d(_p(dst), _p(src), _p(c), int(micro_edges))
    return dst


def p1_elementwise_diagonal_macro_3d(diag, cell_vertex_coords, micro_edges):
    c = np.ascontiguousarray(cell_vertex_coords, dtype=np.float64).reshape(12)
    lib().ho_p1_elementwise_diagonal_macro_3d(_p(diag), _p(c), int(micro_edges))
    return diag


def edge_midpoints(cell_vertex_coords, level):
    """physical coordinates of the edge DoFs (edge midpoints) in array order"""
    cc = np.ascontiguousarray(cell_vertex_coords, dtype=np.float64).reshape(4, 3)
    step = 1.0 / float(1 << level)
    ends = np.array([[[0, 0, 0], [1, 0, 0]], [[0, 0, 0], [0, 1, 0]], [[0, 0, 0], [0, 0, 1]], [[1, 0, 0], [0, 1, 0]],
                     [[1, 0, 0], [0, 0, 1]], [[0, 1, 0], [0, 0, 1]], [[0, 1, 0], [1, 0, 1]]], dtype=np.float64)
    ec = edge_coords(level)
    mid = ec[:, :3].astype(np.float64) + 0.5 * (ends[ec[:, 3], 0] + ends[ec[:, 3], 1])
    xs, ys, zs = (cc[1] - cc[0]) * step, (cc[2] - cc[0]) * step, (cc[3] - cc[0]) * step
    return cc[0][None, :] + mid[:, 0:1] * xs[None, :] + mid[:, 1:2] * ys[None, :] + mid[:, 2:3] * zs[None, :]


def sor_shell_cell(dst, rhs, rest, level, edge_verts, edge_w, face_verts, face_w, vertex_w, relax, mask, backwards=False):
    """vertices -> edges -> faces (reverse if backwards) of one cell; see ho_sor_shell_cell"""
    ev = np.ascontiguousarray(edge_verts, dtype=np.int32).reshape(6, 2)
    fv = np.ascontiguousarray(face_verts, dtype=np.int32).reshape(4, 3)
    ew = np.ascontiguousarray(edge_w, dtype=np.float64).reshape(6, 3)
    fw = np.ascontiguousarray(face_w, dtype=np.float64).reshape(4, 7)
    vw = np.ascontiguousarray(vertex_w, dtype=np.float64).reshape(4)
    lib().ho_sor_shell_cell(_p(dst), _p(rhs), _p(rest), level, ev.ctypes.data_as(C.POINTER(C.c_int)), _p(ew.reshape(-1)),
                            fv.ctypes.data_as(C.POINTER(C.c_int)), _p(fw.reshape(-1)), _p(vw), float(relax), int(mask),
                            1 if backwards else 0)
    return dst


@functools.lru_cache(maxsize=None)
def slot_of_points(level):
    """per array entry: slot 0..13 of the macro-primitive it lies on, 14 for interior points (cached: treat as read-only)"""
    c = cell_coords(level)
    return np.array([14 if prim_slot(level, *map(int, p)) < 0 else prim_slot(level, *map(int, p)) for p in c])


def coordinate_from_index(cell_vertex_coords, level, x, y, z):
    c = np.ascontiguousarray(cell_vertex_coords, dtype=np.float64).reshape(12)
    out = np.empty(3)
    lib().ho_coordinate_from_index(_p(out), _p(c), level, x, y, z)
    return out


def interpolate(cell_vertex_coords, level, fn):
    """Evaluate fn(x,y,z) (vectorised over arrays) at every micro-vertex of the cell array."""
    cc = np.asarray(cell_vertex_coords, dtype=np.float64).reshape(4, 3)
    ijk = cell_coords(level).astype(np.float64)
    step = 1.0 / float(1 << level)
    xs, ys, zs = (cc[1] - cc[0]) * step, (cc[2] - cc[0]) * step, (cc[3] - cc[0]) * step
    p = cc[0][None, :] + ijk[:, 0:1] * xs[None, :] + ijk[:, 1:2] * ys[None, :] + ijk[:, 2:3] * zs[None, :]
    return np.ascontiguousarray(fn(p[:, 0], p[:, 1], p[:, 2]), dtype=np.float64)


# ---- the reference's own FEniCS element matrices, compiled in place (oracle/_ref) ---------------
def ref_fenics():
    """ctypes handle on oracle/_ref/libhyteg_ref_fenics.so or None if it was never built
    (it can only be built where /root/reference is mounted)."""
    so = _HERE / "_ref" / "libhyteg_ref_fenics.so"
    if not so.exists():
        if Path("/root/reference/src/hyteg/forms/form_fenics_generated").is_dir():
            subprocess.run(["make", "-C", str(_HERE), "ref"], check=True, capture_output=True)
        if not so.exists():
            return None
    l = C.CDLL(str(so))
    for n in ("ref_p1_tet_diffusion", "ref_p1_tet_mass", "ref_p2_tet_diffusion", "ref_p1_tet_pspg"):
        if not hasattr(l, n):
            continue
        getattr(l, n).restype, getattr(l, n).argtypes = None, [_P, _P]
    for n in ("ref_p1_tet_div", "ref_p1_tet_divt", "ref_p2_to_p1_tet_div", "ref_p1_to_p2_tet_divt"):
        if hasattr(l, n):
            getattr(l, n).restype, getattr(l, n).argtypes = None, [_P, _P, C.c_int]
    return l


def ref_element_matrix(ref, coords, form):
    """the reference's generated FEniCS tabulate_tensor for form id `form`, row-major A[test i][trial j]"""
    c = np.ascontiguousarray(coords, dtype=np.float64).reshape(12)
    A = np.zeros(16)
    if form == FORM_LAPLACE:
        ref.ref_p1_tet_diffusion(_p(A), _p(c))
    elif form == FORM_MASS:
        ref.ref_p1_tet_mass(_p(A), _p(c))
    elif FORM_DIV_X <= form <= FORM_DIV_Z:
        ref.ref_p1_tet_div(_p(A), _p(c), form - FORM_DIV_X)
    elif FORM_DIVT_X <= form <= FORM_DIVT_Z:
        ref.ref_p1_tet_divt(_p(A), _p(c), form - FORM_DIVT_X)
    else:
        ref.ref_p1_tet_pspg(_p(A), _p(c))
    return A.reshape(4, 4)


def ref_taylor_hood_block(ref, coords, which, k):
    """the reference's FEniCS element matrix of a mixed Taylor-Hood block, padded to 10 x 10 in FEniCS P2 ordering:
    which 0: p2_to_p1_tet_div_tet_cell_integral_k (4 x 10, rows = P1 test functions), 1: p1_to_p2_tet_divt_tet_cell_integral_k (10 x 4)"""
    c = np.ascontiguousarray(coords, dtype=np.float64).reshape(12)
    A = np.zeros(40)
    (ref.ref_p2_to_p1_tet_div if which == 0 else ref.ref_p1_to_p2_tet_divt)(_p(A), _p(c), k)
    M = np.zeros((10, 10))
    if which == 0:
        M[:4, :] = A.reshape(4, 10)
    else:
        M[:, :4] = A.reshape(10, 4)
    return M
