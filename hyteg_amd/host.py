"""ctypes binding of libhyteg_host.so (include/hyteg_host.h): the C facade of the C++ host layer
(hyteg_amd/host/hyteg_host.hpp) that mirrors HyTeG's PrimitiveStorage / P1Function / P1ConstantLaplaceOperator /
grid transfer / solver classes.  Used by the tests, bench.py and the torch.distributed driver; plumbing only."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
_LIB_PATH = _PKG / "lib" / "libhyteg_host.so"

Inner, DirichletBoundary, NeumannBoundary, FreeslipBoundary, All, Boundary = 1, 2, 4, 8, 15, 14
Replace, Add = 0, 1
JACOBI, GAUSS_SEIDEL, SOR, JACOBI_FP32 = 0, 1, 2, 3  # JACOBI_FP32: MixedPrecisionJacobiSmoother (float sweeps on cell interiors)

_vp, _i, _d, _u = C.c_void_p, C.c_int, C.c_double, C.c_uint
_ip, _dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
EXCHANGE_CB = C.CFUNCTYPE(_i, _vp, _i, _i)
ALLREDUCE_CB = C.CFUNCTYPE(_i, _vp, _dp, _i)

SIGNATURES = {
    "hyteg_host_last_error": (C.c_char_p, []),
    "hyteg_host_storage_from_gmsh": (_i, [C.c_char_p, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_storage_from_arrays": (_i, [_i, _dp, _i, _ip, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_storage_destroy": (_i, [_vp]),
    "hyteg_host_storage_counts": (_i, [_vp, _ip]),
    "hyteg_host_storage_local_cell": (_i, [_vp, _i, _ip, _dp, _dp]),
    "hyteg_host_storage_mask": (_i, [_vp, _i, _i, _i, C.POINTER(_u)]),
    "hyteg_host_storage_set_boundary_type": (_i, [_vp, _i]),
    "hyteg_host_storage_set_stream": (_i, [_vp, _vp]),
    "hyteg_host_storage_set_batch_max_level": (_i, [_vp, _i]),
    "hyteg_host_storage_set_hooks": (_i, [_vp, EXCHANGE_CB, EXCHANGE_CB, ALLREDUCE_CB, _vp]),
    "hyteg_host_storage_use_rccl": (_i, [_vp, C.c_char_p]),
    "hyteg_host_storage_use_p2p": (_i, [_vp, C.c_size_t, C.c_char_p, C.POINTER(C.c_int)]),
    "hyteg_host_storage_p2p_open": (_i, [_vp, C.c_char_p]),
    "hyteg_host_storage_p2p_layout": (_i, [_vp, _i, _i, C.POINTER(C.c_longlong)]),
    "hyteg_host_storage_p2p_connect": (_i, [_vp, _i, _i, C.POINTER(C.c_longlong)]),
    "hyteg_host_storage_drop_p2p": (_i, [_vp]),
    "hyteg_host_storage_check_transport": (_i, [_vp]),
    "hyteg_host_storage_transport_name": (_i, [_vp, C.c_char_p, _i]),
    "hyteg_host_storage_allreduce_sum": (_i, [_vp, _dp, _i]),
    "hyteg_host_storage_enable_timing": (_i, [_vp, _i, _i]),
    "hyteg_host_storage_timing_json": (_i, [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "hyteg_host_storage_timing_reset": (_i, [_vp]),
    "hyteg_host_plan_sizes": (_i, [_vp, _i, _i, _ip]),
    "hyteg_host_plan_export": (_i, [_vp, _i, _i, _ip, _ip, _ip, _ip, _ip, _ip, _ip, _ip]),
    "hyteg_host_plan_register_buffers": (_i, [_vp, _i, _i, _vp, _vp]),
    "hyteg_host_function_create": (_i, [_vp, C.c_char_p, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_function_destroy": (_i, [_vp]),
    "hyteg_host_function_cell_pointer": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_function_upload_cell": (_i, [_vp, _i, _i, _dp]),
    "hyteg_host_function_download_cell": (_i, [_vp, _i, _i, _dp]),
    "hyteg_host_function_interpolate_constant": (_i, [_vp, _d, _i, _i]),
    "hyteg_host_function_assign": (_i, [_vp, _i, _dp, C.POINTER(_vp), _i, _i]),
    "hyteg_host_function_add": (_i, [_vp, _i, _dp, C.POINTER(_vp), _i, _i]),
    "hyteg_host_function_mult_elementwise": (_i, [_vp, _i, C.POINTER(_vp), _i, _i]),
    "hyteg_host_function_dot": (_i, [_vp, _vp, _i, _i, _i, _dp]),
    "hyteg_host_function_sum_shared": (_i, [_vp, _i, _i]),
    "hyteg_host_function_sync_shared": (_i, [_vp, _i, _i]),
    "hyteg_host_function_set_all_inner": (_i, [_vp, _i]),
    "hyteg_host_th_function_create": (_i, [_vp, C.c_char_p, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_th_function_destroy": (_i, [_vp]),
    "hyteg_host_th_function_velocity": (_i, [_vp, _i, C.POINTER(_vp)]),
    "hyteg_host_th_function_pressure": (_i, [_vp, C.POINTER(_vp)]),
    "hyteg_host_th_function_assign": (_i, [_vp, _i, _dp, C.POINTER(_vp), _i, _i]),
    "hyteg_host_th_function_interpolate_constant": (_i, [_vp, _d, _i, _i]),
    "hyteg_host_th_function_dot": (_i, [_vp, _vp, _i, _i, C.POINTER(_d)]),
    "hyteg_host_th_operator_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_th_operator_destroy": (_i, [_vp]),
    "hyteg_host_th_operator_apply": (_i, [_vp, _vp, _vp, _i, _i]),
    "hyteg_host_th_operator_apply_block": (_i, [_vp, _i, _vp, _vp, _i, _i]),
    "hyteg_host_th_gmg_create": (_i, [_vp, _i, _i, _d, _i, _i, _i, _i, _d, C.POINTER(_vp)]),
    "hyteg_host_th_minres_create": (_i, [_vp, _i, _i, _i, _d, C.POINTER(_vp)]),
    "hyteg_host_th_solver_solve": (_i, [_vp, _vp, _vp, _vp, _i]),
    "hyteg_host_th_solver_destroy": (_i, [_vp]),
    "hyteg_host_th_project_pressure_mean": (_i, [_vp, _i]),
    "hyteg_host_th_form_element_matrix": (_i, [_i, _i, _dp, _dp]),
    "hyteg_host_stokes_function_create": (_i, [_vp, C.c_char_p, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_stokes_function_destroy": (_i, [_vp]),
    "hyteg_host_stokes_function_component": (_i, [_vp, _i, C.POINTER(_vp)]),
    "hyteg_host_stokes_function_assign": (_i, [_vp, _i, _dp, C.POINTER(_vp), _i, _i]),
    "hyteg_host_stokes_function_dot": (_i, [_vp, _vp, _i, _i, _dp]),
    "hyteg_host_project_mean": (_i, [_vp, _i]),
    "hyteg_host_stokes_operator_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_stokes_operator_destroy": (_i, [_vp]),
    "hyteg_host_stokes_operator_apply": (_i, [_vp, _vp, _vp, _i, _i]),
    "hyteg_host_stokes_uzawa_create": (_i, [_vp, _i, _i, _d, _i, _i, _d, C.POINTER(_vp)]),
    "hyteg_host_stokes_gmg_create": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_stokes_gmg_create_with_coarse": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _d, C.POINTER(_vp)]),
    "hyteg_host_stokes_minres_create": (_i, [_vp, _i, _i, _i, _d, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_stokes_minres_iterations": (_i, [_vp, C.POINTER(_i)]),
    "hyteg_host_solver_create_minres": (_i, [_vp, _i, _i, _i, _d, _i, C.POINTER(_vp)]),
    "hyteg_host_stokes_solver_solve": (_i, [_vp, _vp, _vp, _vp, _i]),
    "hyteg_host_stokes_solver_destroy": (_i, [_vp]),
    "hyteg_host_operator_create": (_i, [_vp, _i, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_operator_destroy": (_i, [_vp]),
    "hyteg_host_operator_stencils": (_i, [_vp, _i, _i, _dp, _dp]),
    "hyteg_host_operator_apply": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "hyteg_host_operator_apply_cycle": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "hyteg_host_operator_apply_cycle_timed": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _i, _i, _vp, _vp]),
    "hyteg_host_operator_smooth_jac": (_i, [_vp, _vp, _vp, _vp, _d, _i, _i]),
    "hyteg_host_operator_smooth_sor": (_i, [_vp, _vp, _vp, _d, _i, _i, _i]),
    "hyteg_host_operator_compute_inverse_diagonal": (_i, [_vp]),
    "hyteg_host_operator_inverse_diagonal": (_i, [_vp, C.POINTER(_vp)]),
    "hyteg_host_elementwise_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_elementwise_destroy": (_i, [_vp]),
    "hyteg_host_elementwise_apply": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "hyteg_host_elementwise_compute_inverse_diagonal": (_i, [_vp]),
    "hyteg_host_elementwise_inverse_diagonal": (_i, [_vp, C.POINTER(_vp)]),
    "hyteg_host_elementwise_smooth_jac": (_i, [_vp, _vp, _vp, _vp, _d, _i, _i]),
    "hyteg_host_restrict": (_i, [_vp, _i, _i]),
    "hyteg_host_prolongate": (_i, [_vp, _i, _i]),
    "hyteg_host_prolongate_and_add": (_i, [_vp, _i, _i]),
    "hyteg_host_gmg_create": (_i, [_vp, _i, _i, _i, _d, _i, _i, _i, _i, _d, C.POINTER(_vp)]),
    "hyteg_host_gmg_set_use_graphs": (_i, [_vp, _i]),
    "hyteg_host_cg_set_use_device_scalars": (_i, [_vp, _i]),
    "hyteg_host_cg_iterations": (_i, [_vp, C.POINTER(_i)]),
    "hyteg_host_gmg_replayed_cycles": (_i, [_vp, C.POINTER(_i)]),
    "hyteg_host_cg_create": (_i, [_vp, _i, _i, _i, _d, C.POINTER(_vp)]),
    "hyteg_host_solver_solve": (_i, [_vp, _vp, _vp, _vp, _i]),
    "hyteg_host_solver_destroy": (_i, [_vp]),
    "hyteg_host_p2function_create": (_i, [_vp, C.c_char_p, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_p2function_destroy": (_i, [_vp]),
    "hyteg_host_p2function_pointers": (_i, [_vp, _i, _i, C.POINTER(_vp), C.POINTER(_vp)]),
    "hyteg_host_p2function_upload": (_i, [_vp, _i, _i, _vp, _vp]),
    "hyteg_host_p2function_download": (_i, [_vp, _i, _i, _vp, _vp]),
    "hyteg_host_p2function_interpolate_constant": (_i, [_vp, _d, _i, _i]),
    "hyteg_host_p2function_assign": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(_vp), _i, _i]),
    "hyteg_host_p2function_add": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(_vp), _i, _i]),
    "hyteg_host_p2function_dot": (_i, [_vp, _vp, _i, _i, C.POINTER(_d)]),
    "hyteg_host_p2operator_create_constant": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_p2_prolongate": (_i, [_vp, _i, _i, _i]),
    "hyteg_host_p2_restrict": (_i, [_vp, _i, _i]),
    "hyteg_host_p2operator_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "hyteg_host_p2operator_destroy": (_i, [_vp]),
    "hyteg_host_p2operator_constant_stencils": (_i, [_vp, _i, _i, _dp, _i, C.POINTER(_i)]),
    "hyteg_host_p2operator_element_matrices": (_i, [_vp, _i, _i, _vp]),
    "hyteg_host_p2operator_apply": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "hyteg_host_p2_cg_solve": (_i, [_vp, _vp, _vp, _vp, _i, _i, _d, C.POINTER(_i)]),
    "hyteg_host_p2operator_compute_inverse_diagonal": (_i, [_vp]),
    "hyteg_host_p2operator_inverse_diagonal_copy": (_i, [_vp, _vp, _i]),
    "hyteg_host_p2operator_smooth_jac": (_i, [_vp, _vp, _vp, _vp, _d, _i, _i]),
    "hyteg_host_p2operator_smooth_sor": (_i, [_vp, _vp, _vp, _d, _i, _i, _i]),
    "hyteg_host_p2_gmg_create": (_i, [_vp, _i, _i, _i, _d, _i, _i, _i, _i, _d, C.POINTER(_vp)]),
    "hyteg_host_p2_solver_solve": (_i, [_vp, _vp, _vp, _vp, _i]),
    "hyteg_host_p2_solver_destroy": (_i, [_vp]),
}


class HytegHostError(RuntimeError):
    pass


_lib = None


def lib_path() -> Path:
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise HytegHostError(f"{_LIB_PATH} not found: run __graft_entry__.build()")
        # libhyteg_host.so depends on libhyteg_hip.so next to it (rpath $ORIGIN)
        l = C.CDLL(str(_LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            f = getattr(l, name)
            f.restype, f.argtypes = res, args
        _lib = l
    return _lib


# an exception raised inside a communication hook (a Python callback called from C++) cannot cross the C frames: the
# trampoline stores it here and returns non-zero, the host layer aborts the operation, and _ck re-raises it
_hook_exception = None


def _ck(rc, what):
    global _hook_exception
    if rc != 0:
        msg = f"{what}: {lib().hyteg_host_last_error().decode(errors='replace')}"
        if _hook_exception is not None:
            exc, _hook_exception = _hook_exception, None
            raise HytegHostError(msg) from exc
        raise HytegHostError(msg)


def cell_size(level: int) -> int:
    n = (1 << level) + 1
    return n * (n + 1) * (n + 2) // 6


class Storage:
    """hyteg::PrimitiveStorage"""

    def __init__(self, handle):
        self.h = handle
        self._hooks = None
        self.stream = 0  # raw hipStream_t the host layer launches on (0: the null stream)
        c = (C.c_int * 6)()
        _ck(lib().hyteg_host_storage_counts(self.h, c), "storage_counts")
        self.n_cells, self.n_faces, self.n_edges, self.n_vertices, self.n_local_cells, self.n_ranks = list(c)

    @classmethod
    def from_gmsh(cls, path, rank=0, nranks=1):
        h = _vp()
        _ck(lib().hyteg_host_storage_from_gmsh(str(path).encode(), rank, nranks, C.byref(h)), "storage_from_gmsh")
        return cls(h)

    @classmethod
    def from_arrays(cls, vertices, cells, rank=0, nranks=1):
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        c = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 4)
        h = _vp()
        _ck(lib().hyteg_host_storage_from_arrays(len(v), v.ctypes.data_as(_dp), len(c), c.ctypes.data_as(_ip), rank, nranks,
                                                 C.byref(h)), "storage_from_arrays")
        return cls(h)

    @classmethod
    def single_tet(cls, coords):
        return cls.from_arrays(coords, [[0, 1, 2, 3]])

    def local_cell(self, i):
        gid = C.c_int()
        co = np.empty(12)
        nnc = np.empty(14)
        _ck(lib().hyteg_host_storage_local_cell(self.h, i, C.byref(gid), co.ctypes.data_as(_dp), nnc.ctypes.data_as(_dp)),
            "storage_local_cell")
        return gid.value, co.reshape(4, 3), nnc

    def mask(self, i, flag, owned=False):
        m = _u()
        _ck(lib().hyteg_host_storage_mask(self.h, i, flag, int(owned), C.byref(m)), "storage_mask")
        return m.value

    def set_boundary_type(self, t):
        _ck(lib().hyteg_host_storage_set_boundary_type(self.h, t), "set_boundary_type")

    def set_batch_max_level(self, level):
        """levels <= level run the batched kernels (one launch for all local cells); -1: per-cell kernels everywhere"""
        _ck(lib().hyteg_host_storage_set_batch_max_level(self.h, level), "set_batch_max_level")

    def set_stream(self, stream):
        _ck(lib().hyteg_host_storage_set_stream(self.h, stream), "set_stream")
        self.stream = int(stream) if stream else 0

    def set_hooks(self, exchange_begin, exchange_end, allreduce_sum):
        """hooks(level, key) with key = cls + 2 * dof_kind; an exception in a hook fails the operation that called it"""

        def guard(fn):
            def call(user, *args):
                global _hook_exception
                try:
                    fn(*args)
                    return 0
                except BaseException as e:  # noqa: BLE001 - must not propagate into the C frames
                    _hook_exception = e
                    return 1
            return call

        exb, exe, ar = EXCHANGE_CB(guard(exchange_begin)), EXCHANGE_CB(guard(exchange_end)), ALLREDUCE_CB(guard(allreduce_sum))
        self._hooks = (exb, exe, ar)  # keep alive
        _ck(lib().hyteg_host_storage_set_hooks(self.h, exb, exe, ar, None), "set_hooks")

    def use_rccl(self, unique_id: bytes):
        """RCCL over xGMI issued from the C++ host layer (collective over all ranks; the rank's device must be current).
        unique_id: the bytes of capi.comm_unique_id() from rank 0, distributed by the caller."""
        if len(unique_id) != 128:
            raise ValueError("use_rccl: the unique id has 128 bytes")
        _ck(lib().hyteg_host_storage_use_rccl(self.h, bytes(unique_id)), "storage_use_rccl")

    # ---- peer-to-peer transport on top of the current one (see include/hyteg_host.h for the set-up sequence) ----
    def use_p2p(self, arena_bytes: int):
        """-> (this rank's arena handle: 64 bytes, arena kind 0 uncached / 1 fine-grained / 2 default)"""
        buf = C.create_string_buffer(64)
        kind = C.c_int(0)
        _ck(lib().hyteg_host_storage_use_p2p(self.h, int(arena_bytes), buf, C.byref(kind)), "storage_use_p2p")
        return buf.raw, kind.value

    def p2p_open(self, handles):
        """handles: the 64-byte handles of all ranks, rank by rank"""
        blob = b"".join(bytes(h) for h in handles)
        _ck(lib().hyteg_host_storage_p2p_open(self.h, blob), "storage_p2p_open")

    def p2p_layout(self, level: int, key: int, npeers: int):
        """-> int64 array [npeers, 3]: byte offsets (slot 0, slot 1, flag) in this rank's arena of every peer's segment"""
        o = np.zeros((max(npeers, 1), 3), dtype=np.int64)
        _ck(lib().hyteg_host_storage_p2p_layout(self.h, level, key, o.ctypes.data_as(C.POINTER(C.c_longlong))), "storage_p2p_layout")
        return o[:npeers]

    def p2p_connect(self, level: int, key: int, offsets):
        o = np.ascontiguousarray(offsets, dtype=np.int64).reshape(-1, 3)
        o = o if len(o) else np.zeros((1, 3), dtype=np.int64)
        _ck(lib().hyteg_host_storage_p2p_connect(self.h, level, key, o.ctypes.data_as(C.POINTER(C.c_longlong))), "storage_p2p_connect")

    def drop_p2p(self):
        _ck(lib().hyteg_host_storage_drop_p2p(self.h), "storage_drop_p2p")

    def check_transport(self):
        """synchronises; raises if a device-side wait of the transport has timed out since the last check"""
        _ck(lib().hyteg_host_storage_check_transport(self.h), "storage_check_transport")

    def allreduce_sum(self, values):
        """sum over all ranks of a sequence of floats (through the storage's transport)"""
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        _ck(lib().hyteg_host_storage_allreduce_sum(self.h, a.ctypes.data_as(_dp), len(a)), "storage_allreduce_sum")
        return a

    def enable_timing(self, on=True, synchronize=False):
        """walberla-style timing tree with the reference's timer names; synchronize: ranges measure device execution"""
        _ck(lib().hyteg_host_storage_enable_timing(self.h, int(on), int(synchronize)), "storage_enable_timing")

    def timing_json(self) -> str:
        """the tree in the layout of walberla::timing::to_json (hyteg::writeTimingTreeJSON)"""
        need = C.c_size_t()
        _ck(lib().hyteg_host_storage_timing_json(self.h, None, 0, C.byref(need)), "storage_timing_json")
        buf = C.create_string_buffer(need.value)
        _ck(lib().hyteg_host_storage_timing_json(self.h, buf, need.value, None), "storage_timing_json")
        return buf.value.decode()

    def timing_reset(self):
        _ck(lib().hyteg_host_storage_timing_reset(self.h), "storage_timing_reset")

    @property
    def transport(self) -> str:
        buf = C.create_string_buffer(32)
        _ck(lib().hyteg_host_storage_transport_name(self.h, buf, 32), "storage_transport_name")
        return buf.value.decode()

    def plan(self, level, key):
        """exchange plan of (level, key = cls + 2 * dof_kind)"""
        cls = key
        s = (C.c_int * 5)()
        _ck(lib().hyteg_host_plan_sizes(self.h, level, cls, s), "plan_sizes")
        ng, ne, npeer, ts, tr = list(s)
        a = lambda n: np.zeros(max(n, 1), dtype=np.int32)  # noqa: E731
        gp, eb, eo, peers, sc, rc, sb, so = a(ng + 1), a(ne), a(ne), a(npeer), a(npeer), a(npeer), a(ts), a(ts)
        p = lambda x: x.ctypes.data_as(_ip)  # noqa: E731
        _ck(lib().hyteg_host_plan_export(self.h, level, cls, p(gp), p(eb), p(eo), p(peers), p(sc), p(rc), p(sb), p(so)),
            "plan_export")
        return dict(ngroups=ng, group_ptr=gp[:ng + 1], entry_buf=eb[:ne], entry_off=eo[:ne], peers=peers[:npeer],
                    send_count=sc[:npeer], recv_count=rc[:npeer], send_buf=sb[:ts], send_off=so[:ts], total_send=ts,
                    total_recv=tr)

    def register_comm_buffers(self, level, cls, send_ptr, recv_ptr):
        _ck(lib().hyteg_host_plan_register_buffers(self.h, level, cls, send_ptr, recv_ptr), "plan_register_buffers")

    def close(self):
        if self.h:
            lib().hyteg_host_storage_destroy(self.h)
            self.h = None


class P1Function:
    """hyteg::P1Function<double>"""

    def __init__(self, storage: Storage, name: str, min_level: int, max_level: int, _borrowed=None):
        self.storage, self.min_level, self.max_level = storage, min_level, max_level
        self._borrowed = _borrowed is not None
        if _borrowed is not None:
            self.h = _borrowed
        else:
            h = _vp()
            _ck(lib().hyteg_host_function_create(storage.h, name.encode(), min_level, max_level, C.byref(h)), "function_create")
            self.h = h

    def cell_pointer(self, c, level):
        p = _vp()
        _ck(lib().hyteg_host_function_cell_pointer(self.h, c, level, C.byref(p)), "cell_pointer")
        return p.value

    def upload_cell(self, c, level, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.size == cell_size(level)
        _ck(lib().hyteg_host_function_upload_cell(self.h, c, level, a.ctypes.data_as(_dp)), "upload_cell")

    def download_cell(self, c, level):
        a = np.empty(cell_size(level))
        _ck(lib().hyteg_host_function_download_cell(self.h, c, level, a.ctypes.data_as(_dp)), "download_cell")
        return a

    def interpolate(self, value, level, flag=All):
        _ck(lib().hyteg_host_function_interpolate_constant(self.h, float(value), level, flag), "interpolate")

    def _vec(self, fn, scalars, funcs, level, flag):
        n = len(funcs)
        hs = (_vp * n)(*[f.h for f in funcs])
        if scalars is None:
            _ck(fn(self.h, n, hs, level, flag), "vector op")
        else:
            sc = (C.c_double * n)(*[float(s) for s in scalars])
            _ck(fn(self.h, n, sc, hs, level, flag), "vector op")

    def assign(self, scalars, funcs, level, flag=All):
        self._vec(lib().hyteg_host_function_assign, scalars, funcs, level, flag)

    def add(self, scalars, funcs, level, flag=All):
        self._vec(lib().hyteg_host_function_add, scalars, funcs, level, flag)

    def mult_elementwise(self, funcs, level, flag=All):
        self._vec(lib().hyteg_host_function_mult_elementwise, None, funcs, level, flag)

    def dot(self, other, level, flag=All, global_=True):
        r = C.c_double()
        _ck(lib().hyteg_host_function_dot(self.h, other.h, level, flag, int(global_), C.byref(r)), "dot")
        return r.value

    def sum_shared(self, level, flag=All):
        _ck(lib().hyteg_host_function_sum_shared(self.h, level, flag), "sum_shared")

    def sync_shared(self, level, flag=All):
        _ck(lib().hyteg_host_function_sync_shared(self.h, level, flag), "sync_shared")

    def set_all_inner(self, on=True):
        """BoundaryCondition::createAllInnerBC(): every point of this function counts as Inner (Stokes pressure)"""
        _ck(lib().hyteg_host_function_set_all_inner(self.h, int(on)), "function_set_all_inner")

    def close(self):
        if self.h and not self._borrowed:
            lib().hyteg_host_function_destroy(self.h)
        self.h = None


FORM_LAPLACE, FORM_MASS, FORM_DIV_X, FORM_DIV_Y, FORM_DIV_Z, FORM_DIVT_X, FORM_DIVT_Y, FORM_DIVT_Z, FORM_PSPG = range(9)


class P1ConstantOperator:
    """hyteg::P1ConstantOperator< Form >: form = FORM_LAPLACE (P1ConstantLaplaceOperator), FORM_MASS, FORM_DIV_X/Y/Z
    (P1Div{x,y,z}Operator), FORM_DIVT_X/Y/Z, FORM_PSPG"""

    def __init__(self, storage: Storage, min_level: int, max_level: int, form: int = 0):
        self.storage = storage
        h = _vp()
        _ck(lib().hyteg_host_operator_create(storage.h, min_level, max_level, form, C.byref(h)), "operator_create")
        self.h = h

    def stencils(self, global_cell, level):
        inner, slots = np.empty(15), np.empty(210)
        _ck(lib().hyteg_host_operator_stencils(self.h, global_cell, level, inner.ctypes.data_as(_dp), slots.ctypes.data_as(_dp)),
            "operator_stencils")
        return inner, slots.reshape(14, 15)

    def apply(self, src, dst, level, flag, update=Replace):
        _ck(lib().hyteg_host_operator_apply(self.h, src.h, dst.h, level, flag, update), "apply")

    def apply_cycle(self, srcs, dsts, level, flag, update=Replace, first=0, steps=1):
        """`steps` applies, step k on pair (first + k) % len(srcs): the loop a C++ application writes around apply()"""
        self.prepared_cycle(srcs, dsts, level, flag, update)(first, steps)

    def prepared_cycle(self, srcs, dsts, level, flag, update=Replace):
        """apply_cycle with the handle arrays built once: returns call(first, steps, ev_start=None, ev_stop=None); the two
        optional timing events of the C-ABI (capi.event_create_timing) are recorded on the storage's stream directly before
        the first and after the last apply"""
        n = len(srcs)
        hs, hd = (_vp * n)(*[f.h for f in srcs]), (_vp * n)(*[f.h for f in dsts])
        fn, fnt, h = lib().hyteg_host_operator_apply_cycle, lib().hyteg_host_operator_apply_cycle_timed, self.h

        def call(first, steps, ev_start=None, ev_stop=None):
            if ev_start is None and ev_stop is None:
                _ck(fn(h, n, hs, hd, level, flag, update, first, steps), "apply_cycle")
            else:
                _ck(fnt(h, n, hs, hd, level, flag, update, first, steps, ev_start, ev_stop), "apply_cycle_timed")

        return call

    def smooth_jac(self, dst, rhs, src, relax, level, flag):
        _ck(lib().hyteg_host_operator_smooth_jac(self.h, dst.h, rhs.h, src.h, float(relax), level, flag), "smooth_jac")

    def smooth_sor(self, dst, rhs, relax, level, flag, backwards=False):
        _ck(lib().hyteg_host_operator_smooth_sor(self.h, dst.h, rhs.h, float(relax), level, flag, int(backwards)), "smooth_sor")

    def smooth_gs(self, dst, rhs, level, flag):
        self.smooth_sor(dst, rhs, 1.0, level, flag)

    def compute_inverse_diagonal(self):
        _ck(lib().hyteg_host_operator_compute_inverse_diagonal(self.h), "computeInverseDiagonalOperatorValues")

    def inverse_diagonal(self, min_level, max_level):
        h = _vp()
        _ck(lib().hyteg_host_operator_inverse_diagonal(self.h, C.byref(h)), "getInverseDiagonalValues")
        return P1Function(self.storage, "invdiag", min_level, max_level, _borrowed=h)

    def close(self):
        if self.h:
            lib().hyteg_host_operator_destroy(self.h)
            self.h = None


class P1ElementwiseDiffusion:
    """hyteg::operatorgeneration::P1ElementwiseDiffusion (the class the hyteg_operators generator emits): every apply goes
    through the C-ABI seam hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_masked with the cell's coordinates"""

    def __init__(self, storage: Storage, min_level: int, max_level: int):
        self.storage = storage
        self.min_level, self.max_level = min_level, max_level
        h = _vp()
        _ck(lib().hyteg_host_elementwise_create(storage.h, min_level, max_level, C.byref(h)), "elementwise_create")
        self.h = h

    def apply(self, src, dst, level, flag, update=Replace):
        _ck(lib().hyteg_host_elementwise_apply(self.h, src.h, dst.h, level, flag, update), "elementwise apply")

    def compute_inverse_diagonal(self):
        _ck(lib().hyteg_host_elementwise_compute_inverse_diagonal(self.h), "computeInverseDiagonalOperatorValues")

    def inverse_diagonal(self):
        h = _vp()
        _ck(lib().hyteg_host_elementwise_inverse_diagonal(self.h, C.byref(h)), "getInverseDiagonalValues")
        return P1Function(self.storage, "invdiag", self.min_level, self.max_level, _borrowed=h)

    def smooth_jac(self, dst, rhs, src, relax, level, flag):
        _ck(lib().hyteg_host_elementwise_smooth_jac(self.h, dst.h, rhs.h, src.h, float(relax), level, flag), "elementwise smooth_jac")

    def close(self):
        if self.h:
            lib().hyteg_host_elementwise_destroy(self.h)
            self.h = None


def restrict(f: P1Function, source_level, flag):
    _ck(lib().hyteg_host_restrict(f.h, source_level, flag), "restrict")


def prolongate(f: P1Function, source_level, flag):
    _ck(lib().hyteg_host_prolongate(f.h, source_level, flag), "prolongate")


def prolongate_and_add(f: P1Function, source_level, flag):
    _ck(lib().hyteg_host_prolongate_and_add(f.h, source_level, flag), "prolongateAndAdd")


class Solver:
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def gmg(cls, storage, min_level, max_level, smoother=JACOBI, relax=2.0 / 3.0, pre=3, post=3, wcycle=False, cg_max_iter=1000,
            cg_tol=1e-14):
        h = _vp()
        _ck(lib().hyteg_host_gmg_create(storage.h, min_level, max_level, smoother, float(relax), pre, post, int(wcycle),
                                        cg_max_iter, float(cg_tol), C.byref(h)), "gmg_create")
        return cls(h)

    @classmethod
    def cg(cls, storage, min_level, max_level, max_iter=1000, tol=1e-14):
        h = _vp()
        _ck(lib().hyteg_host_cg_create(storage.h, min_level, max_level, max_iter, float(tol), C.byref(h)), "cg_create")
        return cls(h)

    @classmethod
    def minres(cls, storage, min_level, max_level, max_iter=1000, rel_tol=1e-16, jacobi_iterations=0):
        """hyteg::MinResSolver< P1ConstantLaplaceOperator >, optionally with JacobiPreconditioner( jacobi_iterations )"""
        h = _vp()
        _ck(lib().hyteg_host_solver_create_minres(storage.h, min_level, max_level, int(max_iter), float(rel_tol), jacobi_iterations, C.byref(h)),
            "solver_create_minres")
        return cls(h)

    def solve(self, laplace: P1ConstantOperator, x: P1Function, b: P1Function, level: int):
        _ck(lib().hyteg_host_solver_solve(self.h, laplace.h, x.h, b.h, level), "solve")

    def set_use_graphs(self, on: bool) -> None:
        """GeometricMultigridSolver::setUseGraphs: record the launches of a cycle once, replay them as graphs (default off)"""
        _ck(lib().hyteg_host_gmg_set_use_graphs(self.h, int(on)), "gmg_set_use_graphs")

    def set_use_device_scalars(self, on: bool, single_launch: bool = False) -> None:
        """CGSolver::setUseDeviceScalars / setUseSingleLaunch (for a multigrid solver: of its coarse-grid CG)"""
        _ck(lib().hyteg_host_cg_set_use_device_scalars(self.h, int(bool(on)) | (2 if single_launch else 0)), "cg_set_use_device_scalars")

    @property
    def iterations(self) -> int:
        n = _i(0)
        _ck(lib().hyteg_host_cg_iterations(self.h, C.byref(n)), "cg_iterations")
        return n.value

    @property
    def replayed_cycles(self) -> int:
        n = _i(0)
        _ck(lib().hyteg_host_gmg_replayed_cycles(self.h, C.byref(n)), "gmg_replayed_cycles")
        return n.value

    def close(self):
        if self.h:
            lib().hyteg_host_solver_destroy(self.h)
            self.h = None


# ---- P2 on a single macro-cell (first version) ----
class P1StokesFunction:
    """hyteg::P1StokesFunction<double>: velocity components u, v, w (boundary types of the storage) and pressure p (all inner)"""

    def __init__(self, storage: Storage, name: str, min_level: int, max_level: int):
        self.storage, self.min_level, self.max_level = storage, min_level, max_level
        h = _vp()
        _ck(lib().hyteg_host_stokes_function_create(storage.h, name.encode(), min_level, max_level, C.byref(h)), "stokes_function_create")
        self.h = h
        self.components = []
        for k in range(4):
            c = _vp()
            _ck(lib().hyteg_host_stokes_function_component(self.h, k, C.byref(c)), "stokes_function_component")
            self.components.append(P1Function(storage, "", min_level, max_level, _borrowed=c))
        self.u, self.v, self.w, self.p = self.components
        self.uvw = self.components[:3]

    def assign(self, scalars, funcs, level, flag=All):
        sc = np.ascontiguousarray(scalars, dtype=np.float64)
        arr = (_vp * len(funcs))(*[f.h for f in funcs])
        _ck(lib().hyteg_host_stokes_function_assign(self.h, len(funcs), sc.ctypes.data_as(_dp), arr, level, flag), "stokes_function_assign")

    def dot(self, other, level, flag=All):
        out = _d()
        _ck(lib().hyteg_host_stokes_function_dot(self.h, other.h, level, flag, C.byref(out)), "stokes_function_dot")
        return out.value

    def close(self):
        if self.h:
            lib().hyteg_host_stokes_function_destroy(self.h)
        self.h = None


def project_mean(pressure: P1Function, level: int):
    """hyteg::vertexdof::projectMean"""
    _ck(lib().hyteg_host_project_mean(pressure.h, level), "project_mean")


class P1P1StokesOperator:
    """hyteg::P1P1StokesOperator: vector Laplace + divT + div + PSPG"""

    def __init__(self, storage: Storage, min_level: int, max_level: int):
        self.storage = storage
        h = _vp()
        _ck(lib().hyteg_host_stokes_operator_create(storage.h, min_level, max_level, C.byref(h)), "stokes_operator_create")
        self.h = h

    def apply(self, src: P1StokesFunction, dst: P1StokesFunction, level, flag):
        _ck(lib().hyteg_host_stokes_operator_apply(self.h, src.h, dst.h, level, flag), "stokes_operator_apply")

    def close(self):
        if self.h:
            lib().hyteg_host_stokes_operator_destroy(self.h)
        self.h = None


class StokesSolver:
    def __init__(self, handle, keep=()):
        self.h = handle
        self._keep = keep

    @classmethod
    def uzawa(cls, storage, min_level, max_level, relax, velocity_iterations=2, velocity_smoother=GAUSS_SEIDEL, velocity_relax=1.0):
        """hyteg::UzawaSmoother< P1P1StokesOperator > over StokesVelocityBlockBlockDiagonalPreconditioner( scalar smoother )"""
        h = _vp()
        _ck(lib().hyteg_host_stokes_uzawa_create(storage.h, min_level, max_level, relax, velocity_iterations, velocity_smoother,
                                                 velocity_relax, C.byref(h)), "stokes_uzawa_create")
        return cls(h)

    @classmethod
    def gmg(cls, storage, smoother, min_level, max_level, pre=3, post=3, increment=0, project_mean_after_restriction=True,
            coarse="lu", coarse_max_iter=1000, coarse_rel_tol=1e-16):
        """hyteg::GeometricMultigridSolver< P1P1StokesOperator >; coarse-grid solver on min_level: "lu" = dense direct solve on
        the host (single rank), "minres" = MinResSolver with the pressure-block preconditioner as apps/stokesSphere composes it
        (any number of ranks)"""
        h = _vp()
        _ck(lib().hyteg_host_stokes_gmg_create_with_coarse(storage.h, smoother.h, min_level, max_level, pre, post, increment,
                                                           int(project_mean_after_restriction), {"lu": 0, "minres": 1}[coarse],
                                                           int(coarse_max_iter), float(coarse_rel_tol), C.byref(h)), "stokes_gmg_create")
        return cls(h, keep=(smoother,))

    @classmethod
    def minres(cls, storage, min_level, max_level, max_iter=1000, rel_tol=1e-16, preconditioner="pressure", velocity_steps=2):
        """hyteg::MinResSolver< P1P1StokesOperator >; preconditioner "identity", "pressure" (StokesPressureBlockPreconditioner with the
        lumped inverse mass operator) or "block" (StokesBlockDiagonalPreconditioner: V(2,2) Laplace cycles on the velocity)"""
        h = _vp()
        _ck(lib().hyteg_host_stokes_minres_create(storage.h, min_level, max_level, int(max_iter), float(rel_tol),
                                                  {"identity": 0, "pressure": 1, "block": 2}[preconditioner], velocity_steps, C.byref(h)),
            "stokes_minres_create")
        return cls(h)

    @property
    def minres_iterations(self) -> int:
        n = _i(0)
        _ck(lib().hyteg_host_stokes_minres_iterations(self.h, C.byref(n)), "stokes_minres_iterations")
        return n.value

    def solve(self, op: P1P1StokesOperator, x: P1StokesFunction, b: P1StokesFunction, level: int):
        _ck(lib().hyteg_host_stokes_solver_solve(self.h, op.h, x.h, b.h, level), "stokes_solver_solve")

    def close(self):
        if self.h:
            lib().hyteg_host_stokes_solver_destroy(self.h)
        self.h = None


class P2Function:
    """hyteg::P2Function<double>: vertex DoFs (the P1 cell array) + edge DoFs (EdgeDoFIndexing.hpp layout)"""

    def __init__(self, storage: Storage, name: str, min_level: int, max_level: int, _borrowed=None):
        self.storage = storage
        self._borrowed = _borrowed is not None
        if _borrowed is not None:
            self.h = _borrowed
            return
        h = _vp()
        _ck(lib().hyteg_host_p2function_create(storage.h, name.encode(), min_level, max_level, C.byref(h)), "P2Function")
        self.h = h

    def sizes(self, level):
        from . import capi

        return cell_size(level), capi.p2_edge_array_size(level)

    def upload(self, level, vertex, edge, cell=0):
        v = np.ascontiguousarray(vertex, dtype=np.float64)
        e = np.ascontiguousarray(edge, dtype=np.float64)
        assert (v.size, e.size) == self.sizes(level)
        _ck(lib().hyteg_host_p2function_upload(self.h, cell, level, v.ctypes.data, e.ctypes.data), "P2Function.upload")

    def download(self, level, cell=0):
        nv, ne = self.sizes(level)
        v, e = np.empty(nv), np.empty(max(ne, 1))
        _ck(lib().hyteg_host_p2function_download(self.h, cell, level, v.ctypes.data, e.ctypes.data), "P2Function.download")
        return v, e[:ne]

    def interpolate(self, value, level, flag=All):
        _ck(lib().hyteg_host_p2function_interpolate_constant(self.h, float(value), level, flag), "P2Function.interpolate")

    def _vec(self, fn, scalars, funcs, level, flag):
        n = len(funcs)
        sc = (_d * n)(*[float(v) for v in scalars])
        hs = (_vp * n)(*[f.h for f in funcs])
        _ck(fn(self.h, n, sc, hs, level, flag), "P2Function vector op")

    def assign(self, scalars, funcs, level, flag=All):
        self._vec(lib().hyteg_host_p2function_assign, scalars, funcs, level, flag)

    def add(self, scalars, funcs, level, flag=All):
        self._vec(lib().hyteg_host_p2function_add, scalars, funcs, level, flag)

    def dot(self, other, level, flag=All):
        r = _d()
        _ck(lib().hyteg_host_p2function_dot(self.h, other.h, level, flag, C.byref(r)), "P2Function.dot")
        return r.value

    def close(self):
        if self.h and not self._borrowed:
            lib().hyteg_host_p2function_destroy(self.h)
        self.h = None


def p2_prolongate(f: "P2Function", source_level, flag, add=False):
    """hyteg::P2toP2QuadraticProlongation::prolongate / prolongateAndAdd"""
    _ck(lib().hyteg_host_p2_prolongate(f.h, source_level, flag, int(add)), "p2_prolongate")


def p2_restrict(f: "P2Function", source_level, flag):
    """hyteg::P2toP2QuadraticRestriction::restrict"""
    _ck(lib().hyteg_host_p2_restrict(f.h, source_level, flag), "p2_restrict")


class P2ElementwiseLaplaceOperator:
    _create = "hyteg_host_p2operator_create"

    def __init__(self, storage: Storage, min_level: int, max_level: int):
        self.storage = storage
        h = _vp()
        _ck(getattr(lib(), self._create)(storage.h, min_level, max_level, C.byref(h)), type(self).__name__)
        self.h = h

    def element_matrices(self, level, cell=0):
        out = np.empty(600)
        _ck(lib().hyteg_host_p2operator_element_matrices(self.h, cell, level, out.ctypes.data), "element_matrices")
        return out.reshape(6, 10, 10)

    def apply(self, src: P2Function, dst: P2Function, level, flag, update=Replace):
        _ck(lib().hyteg_host_p2operator_apply(self.h, src.h, dst.h, level, flag, update), "P2 apply")

    def compute_inverse_diagonal(self):
        _ck(lib().hyteg_host_p2operator_compute_inverse_diagonal(self.h), "P2 computeInverseDiagonalOperatorValues")

    def inverse_diagonal_into(self, dst: "P2Function", level):
        _ck(lib().hyteg_host_p2operator_inverse_diagonal_copy(self.h, dst.h, level), "P2 inverse diagonal")

    def smooth_jac(self, dst: "P2Function", rhs: "P2Function", src: "P2Function", relax, level, flag=Inner | NeumannBoundary):
        _ck(lib().hyteg_host_p2operator_smooth_jac(self.h, dst.h, rhs.h, src.h, float(relax), level, flag), "P2 smooth_jac")

    def smooth_sor(self, dst: "P2Function", rhs: "P2Function", relax, level, flag=Inner | NeumannBoundary, backwards=False):
        _ck(lib().hyteg_host_p2operator_smooth_sor(self.h, dst.h, rhs.h, float(relax), level, flag, int(bool(backwards))), "P2 smooth_sor")

    def smooth_gs(self, dst: "P2Function", rhs: "P2Function", level, flag=Inner | NeumannBoundary):
        self.smooth_sor(dst, rhs, 1.0, level, flag)

    def cg_solve(self, x: P2Function, b: P2Function, level, max_iter=1000, tol=1e-14):
        it = _i()
        _ck(lib().hyteg_host_p2_cg_solve(self.storage.h, self.h, x.h, b.h, level, max_iter, float(tol), C.byref(it)), "P2 CG")
        return it.value

    def close(self):
        if self.h:
            lib().hyteg_host_p2operator_destroy(self.h)
            self.h = None


class P2Solver:
    """GeometricMultigridSolver< P2ElementwiseLaplaceOperator, P2toP2QuadraticRestriction, P2toP2QuadraticProlongation > with a
    weighted-Jacobi smoother and CG on the coarsest level"""

    def __init__(self, storage: Storage, min_level, max_level, relax=2.0 / 3.0, pre=3, post=3, wcycle=False, cg_max_iter=1000, cg_tol=1e-14,
                 smoother=JACOBI):
        self.storage = storage
        h = _vp()
        kind = {JACOBI: 0, GAUSS_SEIDEL: 1, SOR: 2}[smoother]
        _ck(lib().hyteg_host_p2_gmg_create(storage.h, min_level, max_level, kind, float(relax), pre, post, int(bool(wcycle)), cg_max_iter,
                                           float(cg_tol), C.byref(h)), "P2 gmg")
        self.h = h

    def solve(self, op, x: "P2Function", b: "P2Function", level):
        _ck(lib().hyteg_host_p2_solver_solve(self.h, op.h, x.h, b.h, level), "P2 solve")

    def close(self):
        if self.h:
            lib().hyteg_host_p2_solver_destroy(self.h)
            self.h = None


class P2ConstantLaplaceOperator(P2ElementwiseLaplaceOperator):
    """hyteg::P2ConstantLaplaceOperator: assembles the stencils of its four sub-operators itself (P2Elements3D) and hands them to
    the C-ABI's kernel seam; one kernel pass for all four"""
    _create = "hyteg_host_p2operator_create_constant"

    def inner_stencils(self, level, cell=0):
        """values of the v2v | e2v | v2e | e2e maps of a local cell (order: hyteg_hip_p2_constant_stencil_layout)"""
        out, n = np.empty(512), C.c_int(0)
        _ck(lib().hyteg_host_p2operator_constant_stencils(self.h, cell, level, out.ctypes.data_as(_dp), 512, C.byref(n)), "constant_stencils")
        return out[:n.value].copy()


class TaylorHoodFunction:
    """hyteg::P2P1TaylorHoodFunction<double>: three P2 velocity components and a P1 pressure (all-inner boundary condition)"""

    def __init__(self, storage: Storage, name: str, min_level: int, max_level: int):
        self.storage = storage
        h = _vp()
        _ck(lib().hyteg_host_th_function_create(storage.h, name.encode(), min_level, max_level, C.byref(h)), "TaylorHoodFunction")
        self.h = h
        self.velocity = []
        for k in range(3):
            c = _vp()
            _ck(lib().hyteg_host_th_function_velocity(self.h, k, C.byref(c)), "th velocity")
            self.velocity.append(P2Function(storage, "", min_level, max_level, _borrowed=c))
        c = _vp()
        _ck(lib().hyteg_host_th_function_pressure(self.h, C.byref(c)), "th pressure")
        self.pressure = P1Function(storage, "", min_level, max_level, _borrowed=c)

    def assign(self, scalars, funcs, level, flag=All):
        n = len(funcs)
        sc = (_d * n)(*[float(v) for v in scalars])
        hs = (_vp * n)(*[f.h for f in funcs])
        _ck(lib().hyteg_host_th_function_assign(self.h, n, sc, hs, level, flag), "th assign")

    def interpolate(self, value, level, flag=All):
        _ck(lib().hyteg_host_th_function_interpolate_constant(self.h, float(value), level, flag), "th interpolate")

    def dot(self, other, level, flag=All):
        r = _d()
        _ck(lib().hyteg_host_th_function_dot(self.h, other.h, level, flag, C.byref(r)), "th dot")
        return r.value

    def project_pressure_mean(self, level):
        _ck(lib().hyteg_host_th_project_pressure_mean(self.h, level), "th projectMean")

    def close(self):
        if self.h:
            lib().hyteg_host_th_function_destroy(self.h)
            self.h = None


class TaylorHoodStokesOperator:
    """hyteg::P2P1TaylorHoodStokesOperator"""

    def __init__(self, storage: Storage, min_level: int, max_level: int):
        self.storage = storage
        h = _vp()
        _ck(lib().hyteg_host_th_operator_create(storage.h, min_level, max_level, C.byref(h)), "TaylorHoodStokesOperator")
        self.h = h

    def apply(self, src, dst, level, flag):
        _ck(lib().hyteg_host_th_operator_apply(self.h, src.h, dst.h, level, flag), "th apply")

    def apply_div(self, src, dst, level, flag):
        _ck(lib().hyteg_host_th_operator_apply_block(self.h, 0, src.h, dst.h, level, flag), "th div")

    def apply_divt(self, src, dst, level, flag):
        _ck(lib().hyteg_host_th_operator_apply_block(self.h, 1, src.h, dst.h, level, flag), "th divT")

    def close(self):
        if self.h:
            lib().hyteg_host_th_operator_destroy(self.h)
            self.h = None


class TaylorHoodSolver:
    def __init__(self, h):
        self.h = h

    @classmethod
    def gmg(cls, storage, min_level, max_level, uzawa_relax=0.4, pre=3, post=3, increment=0, coarse_max_iter=200, coarse_rel_tol=1e-12):
        h = _vp()
        _ck(lib().hyteg_host_th_gmg_create(storage.h, min_level, max_level, float(uzawa_relax), pre, post, increment, coarse_max_iter,
                                           float(coarse_rel_tol), C.byref(h)), "th gmg")
        return cls(h)

    @classmethod
    def minres(cls, storage, min_level, max_level, max_iter=100, rel_tol=1e-10):
        h = _vp()
        _ck(lib().hyteg_host_th_minres_create(storage.h, min_level, max_level, max_iter, float(rel_tol), C.byref(h)), "th minres")
        return cls(h)

    def solve(self, op, x, b, level):
        _ck(lib().hyteg_host_th_solver_solve(self.h, op.h, x.h, b.h, level), "th solve")

    def close(self):
        if self.h:
            lib().hyteg_host_th_solver_destroy(self.h)
            self.h = None


def taylor_hood_form_element_matrix(which, k, coords):
    """10 x 10 padded element matrix of a mixed block: which 0 = div (P2 -> P1), 1 = divT (P1 -> P2); component k"""
    co = np.ascontiguousarray(coords, dtype=np.float64).reshape(12)
    out = np.empty(100)
    _ck(lib().hyteg_host_th_form_element_matrix(which, k, co.ctypes.data_as(_dp), out.ctypes.data_as(_dp)), "th_form_element_matrix")
    return out.reshape(10, 10)
