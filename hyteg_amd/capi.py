"""ctypes binding of libhyteg_hip.so (include/hyteg_hip.h).  Plumbing only: every call goes straight to
the C-ABI; device memory comes from the caller (e.g. torch CUDA tensors' data_ptr()) or hyteg_hip_malloc."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_LIB_PATH = _PKG / "lib" / "libhyteg_hip.so"

REPLACE, ADD = 0, 1
MIN_LEVEL, MAX_LEVEL, MAX_SRCS = 2, 11, 4

# every symbol include/hyteg_hip.h declares: (restype, argtypes)
_vp, _i, _d, _sz, _i64 = C.c_void_p, C.c_int, C.c_double, C.c_size_t, C.c_int64
_dp = C.POINTER(C.c_double)
SIGNATURES = {
    "hyteg_hip_version": (C.c_char_p, []),
    "hyteg_hip_last_error": (C.c_char_p, []),
    "hyteg_hip_device_count": (_i, [C.POINTER(_i)]),
    "hyteg_hip_set_device": (_i, [_i]),
    "hyteg_hip_device_name": (_i, [C.c_char_p, _sz]),
    "hyteg_hip_malloc": (_i, [C.POINTER(_vp), _sz]),
    "hyteg_hip_free": (_i, [_vp]),
    "hyteg_hip_memset_zero": (_i, [_vp, _sz, _vp]),
    "hyteg_hip_upload": (_i, [_vp, _vp, _sz, _vp]),
    "hyteg_hip_download": (_i, [_vp, _vp, _sz, _vp]),
    "hyteg_hip_copy": (_i, [_vp, _vp, _sz, _vp]),
    "hyteg_hip_calib_copy": (_i, [_vp, _vp, _i64, _i, _vp]),
    "hyteg_hip_p2_operator_table_closure_split": (_i, [_vp, _vp, _vp, _vp]),
    "hyteg_hip_p2_operator_table_face_edge_weights": (_i, [_vp, C.POINTER(_i), _vp]),
    "hyteg_hip_p2_sor_face_frames_bytes": (_sz, []),
    "hyteg_hip_p2_sor_face_frames": (_i, [_i, C.POINTER(_i), _vp, _vp]),
    "hyteg_hip_p2_sor_face_edgedofs_cells": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), _i, _vp, _d, C.POINTER(C.c_uint), _i, _vp]),
    "hyteg_hip_p2_sor_face_edgedofs_cell": (_i, [_vp, _vp, _i, C.POINTER(_i), _vp, _d, C.c_uint, _i, _vp]),
    "hyteg_hip_calib_copy_ring": (_i, [C.POINTER(_vp), C.POINTER(_vp), _i, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "hyteg_hip_stream_create": (_i, [C.POINTER(_vp)]),
    "hyteg_hip_stream_destroy": (_i, [_vp]),
    "hyteg_hip_stream_synchronize": (_i, [_vp]),
    "hyteg_hip_event_create": (_i, [C.POINTER(_vp)]),
    "hyteg_hip_event_create_timing": (_i, [C.POINTER(_vp)]),
    "hyteg_hip_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "hyteg_hip_event_destroy": (_i, [_vp]),
    "hyteg_hip_event_record": (_i, [_vp, _vp]),
    "hyteg_hip_stream_wait_event": (_i, [_vp, _vp]),
    "hyteg_hip_comm_available": (_i, [C.c_char_p, _sz]),
    "hyteg_hip_comm_unique_id": (_i, [C.c_char_p]),
    "hyteg_hip_comm_create": (_i, [C.POINTER(_vp), _i, _i, C.c_char_p]),
    "hyteg_hip_comm_destroy": (_i, [_vp]),
    "hyteg_hip_comm_exchange": (_i, [_vp, _i, C.POINTER(_i), _vp, C.POINTER(_i), _vp, C.POINTER(_i), _vp]),
    "hyteg_hip_comm_allreduce_sum": (_i, [_vp, _vp, _i, _vp]),
    "hyteg_hip_p2p_arena_create": (_i, [C.c_size_t, C.POINTER(_vp), C.c_char_p, C.POINTER(_i)]),
    "hyteg_hip_p2p_arena_destroy": (_i, [_vp]),
    "hyteg_hip_p2p_arena_open": (_i, [C.c_char_p, C.POINTER(_vp)]),
    "hyteg_hip_p2p_arena_close": (_i, [_vp]),
    "hyteg_hip_p2p_pack": (_i, [_vp, _i, _vp, _vp, _vp, _i, C.c_ulonglong, _vp, _vp]),
    "hyteg_hip_p2p_wait": (_i, [_vp, _i, _i, C.c_ulonglong, _vp, C.c_uint, _vp]),
    "hyteg_hip_p1_apply_cell_boundary_p2p": (_i, [_vp, _vp, _i, _vp, C.c_uint, _i, _vp, _vp, _vp, _i, C.c_ulonglong, _vp, _vp]),
    "hyteg_hip_reduce_shared_after_p2p": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, C.c_ulonglong, _vp, C.c_uint, _vp]),
    "hyteg_hip_graph_begin_capture": (_i, [_vp]),
    "hyteg_hip_graph_end_capture": (_i, [_vp, C.POINTER(_vp)]),
    "hyteg_hip_graph_abort_capture": (_i, [_vp]),
    "hyteg_hip_graph_launch": (_i, [_vp, _vp]),
    "hyteg_hip_graph_destroy": (_i, [_vp]),
    "hyteg_hip_prepare_level": (_i, [_i]),
    "hyteg_hip_cell_width": (_i64, [_i]),
    "hyteg_hip_cell_size": (_i64, [_i]),
    "hyteg_hip_cell_inner_size": (_i64, [_i]),
    "hyteg_hip_cell_index": (_i64, [_i, _i, _i, _i]),
    "hyteg_hip_p1_apply_cell": (_i, [_vp, _vp, _i, _dp, _i, _vp]),
    "hyteg_hip_p1_apply_kernel_name": (_i, [_i, _i, C.c_char_p, _sz]),
    "hyteg_hip_p1_residual_jacobi_start_f32": (_i, [_vp, _vp, _vp, _vp, _i, _dp, _d, _vp]),
    "hyteg_hip_p1_jacobi_accumulate_f32": (_i, [_vp, _vp, _vp, _i, _dp, _d, _vp]),
    "hyteg_hip_set_apply_shape": (_i, [_i, _i, _i]),
    "hyteg_hip_p1_apply_cell_f32": (_i, [_vp, _vp, _i, _dp, _i, _vp]),
    "hyteg_hip_p1_jacobi_cell_f32": (_i, [_vp, _vp, _vp, _vp, _i, _dp, _d, _vp]),
    "hyteg_hip_convert_f64_to_f32": (_i, [_vp, _vp, _sz, _vp]),
    "hyteg_hip_convert_f32_to_f64": (_i, [_vp, _vp, _sz, _vp]),
    "hyteg_hip_axpy_f32_into_f64": (_i, [_vp, _vp, _d, _sz, _vp]),
    "hyteg_hip_p1_jacobi_cell": (_i, [_vp, _vp, _vp, _vp, _i, _dp, _d, _vp]),
    "hyteg_hip_p1_sor_cell": (_i, [_vp, _vp, _i, _dp, _d, _i, _vp]),
    "hyteg_hip_p1_residual_cell": (_i, [_vp, _vp, _vp, _i, _dp, _vp]),
    "hyteg_hip_p1_sor_cell_sweeps": (_i, [_vp, _vp, _i, _dp, _d, _i, _i, _vp]),
    "hyteg_hip_set_sor_algorithm": (_i, [_i]),
    "hyteg_hip_p1_assign_cell": (_i, [_vp, _i, C.POINTER(_vp), _dp, _i, _vp]),
    "hyteg_hip_p1_add_cell": (_i, [_vp, _i, C.POINTER(_vp), _dp, _i, _vp]),
    "hyteg_hip_p1_mult_cell": (_i, [_vp, _i, C.POINTER(_vp), _i, _vp]),
    "hyteg_hip_dot_workspace_bytes": (_sz, []),
    "hyteg_hip_p1_dot_cell": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "hyteg_hip_p1_restrict_cell": (_i, [_vp, _vp, _i, _dp, _vp]),
    "hyteg_hip_p1_prolongate_cell": (_i, [_vp, _vp, _i, _dp, _i, _vp]),
    "hyteg_hip_p1_apply_cell_boundary": (_i, [_vp, _vp, _i, _dp, C.c_uint, _i, _vp]),
    "hyteg_hip_p1_vector_cell_masked": (_i, [_i, _vp, _i, C.POINTER(_vp), _dp, _i, C.c_uint, _vp]),
    "hyteg_hip_p1_set_cell_masked": (_i, [_vp, _d, _i, C.c_uint, _vp]),
    "hyteg_hip_p1_dot_cell_masked": (_i, [_vp, _vp, _i, C.c_uint, _vp, _vp, _vp]),
    "hyteg_hip_p1_restrict_cell_masked": (_i, [_vp, _vp, _i, _dp, C.c_uint, _vp]),
    "hyteg_hip_p1_prolongate_cell_masked": (_i, [_vp, _vp, _i, _dp, C.c_uint, _vp]),
    "hyteg_hip_p1_prolongate_cell_masked_update": (_i, [_vp, _vp, _i, _dp, C.c_uint, _i, _vp]),
    "hyteg_hip_sum_shared": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "hyteg_hip_copy_shared": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "hyteg_hip_p1_copy_face_to_cell": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "hyteg_hip_p1_copy_cell_to_face": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "hyteg_hip_p1_vector_cells": (_i, [_i, _i, C.POINTER(_vp), _i, C.POINTER(_vp), _dp, _i, C.POINTER(C.c_uint), _vp]),
    "hyteg_hip_p1_dot_cells": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), _i, C.POINTER(C.c_uint), _vp, _vp, _vp]),
    "hyteg_hip_p1_apply_cells": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), _i, _vp, C.POINTER(C.c_uint), _i, _vp]),
    "hyteg_hip_p1_jacobi_cells": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _vp, _d,
                                        C.POINTER(C.c_uint), _i, _vp]),
    "hyteg_hip_p1_restrict_cells": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), _i, _vp, C.POINTER(C.c_uint), _vp]),
    "hyteg_hip_p1_prolongate_cells": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), _i, _vp, C.POINTER(C.c_uint), _i, _vp]),
    "hyteg_hip_p2_edge_array_size": (C.c_size_t, [_i]),
    "hyteg_hip_p2_operator_table_size": (C.c_size_t, []),
    "hyteg_hip_p2_build_operator_table": (_i, [_dp, _dp]),
    "hyteg_hip_p2_prolongate_cell": (_i, [_vp, _vp, _vp, _vp, _i, _i, C.c_uint, _vp]),
    "hyteg_hip_p2_restrict_cell": (_i, [_vp, _vp, _vp, _vp, _i, _dp, C.c_uint, _vp]),
    "hyteg_hip_p2_edge_vector_cell_masked": (_i, [_i, _vp, _i, C.POINTER(_vp), _dp, _i, C.c_uint, _vp]),
    "hyteg_hip_p2_edge_vector_cell_kinds": (_i, [_i, _vp, _i, C.POINTER(_vp), _dp, _i, C.c_uint, C.c_uint, _vp]),
    "hyteg_hip_p2_elementwise_apply_cells_kinds": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, C.POINTER(_vp), _d, _i,
                                                   C.POINTER(C.c_uint), C.c_uint, _vp]),
    "hyteg_hip_p2_edge_dot_cells_masked": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), _i, C.POINTER(C.c_uint), _vp, _vp]),
    "hyteg_hip_p2_edge_vector_cells_kinds": (_i, [_i, _i, C.POINTER(_vp), _i, C.POINTER(_vp), _dp, _i, C.POINTER(C.c_uint), C.c_uint, _vp]),
    "hyteg_hip_p2_edge_dot_cell_masked": (_i, [_vp, _vp, _i, C.c_uint, _vp, _vp, _vp]),
    "hyteg_hip_p2_elementwise_apply_cell": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _d, _i, C.c_uint, _vp]),
    "hyteg_hip_p2_elementwise_apply_cell_kinds": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _d, _i, C.c_uint, C.c_uint, _vp]),
    "hyteg_hip_p2_set_class_rows_min_level": (_i, [_i]),
    "hyteg_hip_p1_vector_cells_dev": (_i, [_i, _i, C.POINTER(_vp), _i, C.POINTER(_vp), C.POINTER(_vp), _i, C.POINTER(C.c_uint), _vp]),
    "hyteg_hip_cg_scalars": (_i, [_vp, _i, _d, _d, _vp]),
    "hyteg_hip_p1_cg_small_max_entries": (_i, []),
    "hyteg_hip_p1_cg_small_cells": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), _i, _vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(_vp),
                                         C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), _i, _d, _d, _vp, _vp]),
    "hyteg_hip_p1_dot_cells_cg": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), _i, C.POINTER(C.c_uint), _vp, _i, _i, _d, _d, _vp, _vp]),
    "hyteg_hip_p1_sor_cells": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), _i, _vp, _d, _i, C.POINTER(C.c_uint), _vp]),
    "hyteg_hip_p1_sor_shell_cells": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _vp, _d, C.POINTER(C.c_uint), _i, _vp]),
    "hyteg_hip_p1_sor_shell_cell": (_i, [_vp, _vp, _vp, _i, C.POINTER(_i), _dp, C.POINTER(_i), _dp, _dp, _d, C.c_uint, _i, _vp]),
    "hyteg_hip_p1_apply_face3d": (_i, [_vp, _vp, _i, _i, C.POINTER(_i), _dp, _i, _vp]),
    "hyteg_hip_p1_sor_face3d_workspace": (C.c_size_t, [_i]),
    "hyteg_hip_p1_sor_face3d": (_i, [_vp, _vp, _vp, _i, _i, C.POINTER(_i), _dp, C.c_double, _i, _vp]),
    "hyteg_hip_gather_entries": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "hyteg_hip_p2_constant_stencil_layout": (_i, [C.POINTER(_i), C.POINTER(_i)]),
    "hyteg_hip_p2_build_operator_table_from_stencils": (_i, [_dp, _dp, _dp]),
    "hyteg_hip_p2_apply_cell_edgedof_to_vertexdof": (_i, [_vp] * 7 + [_vp, _dp, _i, _i, _vp]),
    "hyteg_hip_p2_apply_cell_vertexdof_to_edgedof": (_i, [_vp] * 7 + [_vp, _i, _dp, _i, _vp]),
    "hyteg_hip_p2_apply_cell_edgedof_to_edgedof": (_i, [_vp] * 14 + [_dp, _i, _i, _vp]),
    "hyteg_hip_p1_elementwise_diffusion_apply_macro_3d": (_i, [_vp, _vp, _dp, _i64, _d, _vp]),
    "hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_masked": (_i, [_vp, _vp, _dp, _i64, C.c_uint, _i, _vp]),
    "hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_f32": (_i, [_vp, _vp, C.POINTER(C.c_float), _i64, C.c_float, _vp]),
    "hyteg_hip_p1_elementwise_diffusion_diagonal_macro_3d": (_i, [_vp, _dp, _i64, _d, _vp]),
    "hyteg_hip_p1_elementwise_diffusion_stencils": (_i, [_dp, _i64, _dp, _dp]),
    "hyteg_hip_p1_apply_cell_boundary_f32": (_i, [_vp, _vp, _i, _dp, C.c_uint, _i, _vp]),
    "hyteg_hip_p2_elementwise_diffusion_apply_macro_3d": (_i, [_vp, _vp, _vp, _vp, _dp, _i64, _d, _vp]),
    "hyteg_hip_p2_elementwise_diffusion_element_matrices": (_i, [_dp, _i64, _dp]),
}
MASK_INNER, MASK_SHELL, MASK_ALL = 1 << 14, 0x3FFF, 0x7FFF


class HytegHipError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """The loaded C-ABI library.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise HytegHipError(
                f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  hyteg_amd has no CPU fallback.")
        l = C.CDLL(str(_LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            f = getattr(l, name)  # AttributeError here = header and library out of sync
            f.restype, f.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().hyteg_hip_last_error().decode(errors="replace")
        raise HytegHipError(f"{what or 'hyteg_hip call'} failed (code {rc}): {msg}")


def _w15(w):
    arr = (C.c_double * 15)(*[float(v) for v in w])
    return arr


def _f14(v):
    return (C.c_double * 14)(*[float(x) for x in v])


def _ptrs(ptrs):
    return (_vp * len(ptrs))(*[int(p) for p in ptrs])


def _scal(s):
    return (C.c_double * len(s))(*[float(x) for x in s])


# ---- thin, checked wrappers (pointers are ints: device addresses) -------------------------------------------
def cell_size(level): return int(lib().hyteg_hip_cell_size(level))
def cell_inner_size(level): return int(lib().hyteg_hip_cell_inner_size(level))
def cell_width(level): return int(lib().hyteg_hip_cell_width(level))
def cell_index(level, x, y, z): return int(lib().hyteg_hip_cell_index(level, x, y, z))
def prepare_level(level): check(lib().hyteg_hip_prepare_level(level), "prepare_level")


def device_name() -> str:
    buf = C.create_string_buffer(256)
    check(lib().hyteg_hip_device_name(buf, 256), "device_name")
    return buf.value.decode()


def comm_available() -> str:
    """where librccl was resolved from (raises if it cannot be)"""
    buf = C.create_string_buffer(256)
    check(lib().hyteg_hip_comm_available(buf, 256), "comm_available")
    return buf.value.decode()


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    check(lib().hyteg_hip_comm_unique_id(buf), "comm_unique_id")
    return buf.raw


def comm_create(nranks: int, rank: int, unique_id: bytes) -> int:
    h = _vp()
    check(lib().hyteg_hip_comm_create(C.byref(h), nranks, rank, bytes(unique_id)), "comm_create")
    return h.value


def comm_destroy(comm) -> None:
    check(lib().hyteg_hip_comm_destroy(comm), "comm_destroy")


def comm_exchange(comm, peers, send_ptr, send_count, recv_ptr, recv_count, stream=0) -> None:
    n = len(peers)
    ia = lambda v: (C.c_int * max(n, 1))(*[int(x) for x in v])  # noqa: E731
    check(lib().hyteg_hip_comm_exchange(comm, n, ia(peers), send_ptr, ia(send_count), recv_ptr, ia(recv_count), stream), "comm_exchange")


def comm_allreduce_sum(comm, dev_ptr, n, stream=0) -> None:
    check(lib().hyteg_hip_comm_allreduce_sum(comm, dev_ptr, n, stream), "comm_allreduce_sum")


def event_create() -> int:
    h = _vp()
    check(lib().hyteg_hip_event_create(C.byref(h)), "event_create")
    return h.value


def event_create_timing() -> int:
    h = _vp()
    check(lib().hyteg_hip_event_create_timing(C.byref(h)), "event_create_timing")
    return h.value


def event_elapsed_ms(start, stop) -> float:
    """waits for `stop`; device time between the two timing events"""
    ms = C.c_float()
    check(lib().hyteg_hip_event_elapsed_ms(start, stop, C.byref(ms)), "event_elapsed_ms")
    return float(ms.value)


def event_destroy(ev) -> None:
    check(lib().hyteg_hip_event_destroy(ev), "event_destroy")


def event_record(ev, stream=0) -> None:
    check(lib().hyteg_hip_event_record(ev, stream), "event_record")


def stream_wait_event(stream, ev) -> None:
    check(lib().hyteg_hip_stream_wait_event(stream, ev), "stream_wait_event")


def calib_copy(dst, src, n, nontemporal=True, stream=0) -> None:
    """streaming copy of n doubles (calibration of the practical bandwidth floor; bench.py roofline.copy_us)"""
    check(lib().hyteg_hip_calib_copy(dst, src, n, 1 if nontemporal else 0, stream), "calib_copy")


def calib_copy_ring(pairs, n, nontemporal=True, stream=0):
    """returns call(first, count, ev_start=None, ev_stop=None): `count` calib copies, copy k on pairs[(first + k) % len(pairs)]
    (pairs of (dst, src) device pointers), issued by one C loop between the two optional timing events"""
    m = len(pairs)
    d, s = (_vp * m)(*[p[0] for p in pairs]), (_vp * m)(*[p[1] for p in pairs])
    fn = lib().hyteg_hip_calib_copy_ring

    def call(first, count, ev_start=None, ev_stop=None):
        check(fn(d, s, m, n, 1 if nontemporal else 0, first, count, stream, ev_start, ev_stop), "calib_copy_ring")

    return call


def p1_residual_jacobi_start_f32(r_f32, e_f32, rhs, src, level, w, relax, stream=0):
    check(lib().hyteg_hip_p1_residual_jacobi_start_f32(r_f32, e_f32, rhs, src, level, _w15(w), float(relax), stream), "p1_residual_jacobi_start_f32")


def p1_jacobi_accumulate_f32(x, rhs_f32, e_f32, level, w, relax, stream=0):
    check(lib().hyteg_hip_p1_jacobi_accumulate_f32(x, rhs_f32, e_f32, level, _w15(w), float(relax), stream), "p1_jacobi_accumulate_f32")


def set_apply_shape(ny=0, lz=0, pfd=0) -> None:
    """tuning knob: brick shape of the z-march kernels (0, 0, 0 = per-level defaults)"""
    check(lib().hyteg_hip_set_apply_shape(ny, lz, pfd), "set_apply_shape")


def p1_apply_kernel_name(level, update=REPLACE) -> str:
    buf = C.create_string_buffer(256)
    check(lib().hyteg_hip_p1_apply_kernel_name(level, update, buf, 256), "p1_apply_kernel_name")
    return buf.value.decode()


def p2_prolongate_cell(fine_v, fine_e, coarse_v, coarse_e, coarse_level, update=REPLACE, mask=0x7FFF, stream=0):
    check(lib().hyteg_hip_p2_prolongate_cell(fine_v, fine_e, coarse_v, coarse_e, coarse_level, update, mask, stream), "p2_prolongate_cell")


def p2_restrict_cell(coarse_v, coarse_e, fine_v, fine_e, coarse_level, nnc, mask=0x7FFF, stream=0):
    a = (C.c_double * 14)(*[float(x) for x in nnc])
    check(lib().hyteg_hip_p2_restrict_cell(coarse_v, coarse_e, fine_v, fine_e, coarse_level, a, mask, stream), "p2_restrict_cell")


def p1_apply_cell_f32(dst, src, level, w, update=REPLACE, stream=0):
    check(lib().hyteg_hip_p1_apply_cell_f32(dst, src, level, _w15(w), update, stream), "p1_apply_cell_f32")


def p1_jacobi_cell_f32(dst, rhs, src, level, w, relax, invdiag=None, stream=0):
    check(lib().hyteg_hip_p1_jacobi_cell_f32(dst, rhs, src, invdiag, level, _w15(w), relax, stream), "p1_jacobi_cell_f32")


def convert_f64_to_f32(dst, src, n, stream=0):
    check(lib().hyteg_hip_convert_f64_to_f32(dst, src, n, stream), "convert_f64_to_f32")


def convert_f32_to_f64(dst, src, n, stream=0):
    check(lib().hyteg_hip_convert_f32_to_f64(dst, src, n, stream), "convert_f32_to_f64")


def axpy_f32_into_f64(y, x, alpha, n, stream=0):
    check(lib().hyteg_hip_axpy_f32_into_f64(y, x, alpha, n, stream), "axpy_f32_into_f64")


def p1_apply_cell(dst, src, level, w, update=REPLACE, stream=0):
    check(lib().hyteg_hip_p1_apply_cell(dst, src, level, _w15(w), update, stream), "p1_apply_cell")


def p1_jacobi_cell(dst, rhs, src, level, w, relax, invdiag=None, stream=0):
    check(lib().hyteg_hip_p1_jacobi_cell(dst, rhs, src, invdiag, level, _w15(w), float(relax), stream),
          "p1_jacobi_cell")


SOR_AUTO, SOR_PLANES, SOR_BLOCKS, SOR_DATAFLOW = 0, 1, 2, 3


def set_sor_algorithm(algorithm: int) -> None:
    """hyteg_hip_set_sor_algorithm: how the cell sweeps are executed (tests / benchmarks; default SOR_AUTO)"""
    check(lib().hyteg_hip_set_sor_algorithm(int(algorithm)), "set_sor_algorithm")


def p1_sor_cell(u, rhs, level, w, relax, backwards=False, stream=0):
    check(lib().hyteg_hip_p1_sor_cell(u, rhs, level, _w15(w), float(relax), int(backwards), stream), "p1_sor_cell")


def p1_residual_cell(dst, rhs, src, level, w, stream=0):
    check(lib().hyteg_hip_p1_residual_cell(dst, rhs, src, level, _w15(w), stream), "p1_residual_cell")


def p1_sor_cell_sweeps(u, rhs, level, w, relax, nsweeps, backwards=False, stream=0):
    check(lib().hyteg_hip_p1_sor_cell_sweeps(u, rhs, level, _w15(w), float(relax), int(backwards), int(nsweeps), stream), "p1_sor_cell_sweeps")


def p1_assign_cell(dst, scalars, srcs, level, stream=0):
    check(lib().hyteg_hip_p1_assign_cell(dst, len(srcs), _ptrs(srcs), _scal(scalars), level, stream),
          "p1_assign_cell")


def p1_add_cell(dst, scalars, srcs, level, stream=0):
    check(lib().hyteg_hip_p1_add_cell(dst, len(srcs), _ptrs(srcs), _scal(scalars), level, stream), "p1_add_cell")


def p1_mult_cell(dst, srcs, level, stream=0):
    check(lib().hyteg_hip_p1_mult_cell(dst, len(srcs), _ptrs(srcs), level, stream), "p1_mult_cell")


def dot_workspace_bytes(): return int(lib().hyteg_hip_dot_workspace_bytes())


def p1_dot_cell(a, b, level, result_dev, workspace_dev, stream=0):
    check(lib().hyteg_hip_p1_dot_cell(a, b, level, result_dev, workspace_dev, stream), "p1_dot_cell")


def p1_restrict_cell(coarse, fine, coarse_level, nnc, stream=0):
    check(lib().hyteg_hip_p1_restrict_cell(coarse, fine, coarse_level, _f14(nnc), stream), "p1_restrict_cell")


def p1_prolongate_cell(coarse, fine, coarse_level, nnc, update=REPLACE, stream=0):
    check(lib().hyteg_hip_p1_prolongate_cell(coarse, fine, coarse_level, _f14(nnc), update, stream),
          "p1_prolongate_cell")


def p1_apply_cell_boundary(dst, src, level, w_slots, mask=MASK_SHELL, update=REPLACE, stream=0):
    flat = [float(v) for row in w_slots for v in row]
    arr = (C.c_double * 210)(*flat)
    check(lib().hyteg_hip_p1_apply_cell_boundary(dst, src, level, arr, mask, update, stream), "p1_apply_cell_boundary")


def p1_vector_cell_masked(op, dst, scalars, srcs, level, mask, stream=0):
    sc = _scal(scalars) if scalars is not None else None
    check(lib().hyteg_hip_p1_vector_cell_masked(op, dst, len(srcs), _ptrs(srcs), sc, level, mask, stream),
          "p1_vector_cell_masked")


def p1_set_cell_masked(dst, value, level, mask, stream=0):
    check(lib().hyteg_hip_p1_set_cell_masked(dst, float(value), level, mask, stream), "p1_set_cell_masked")


def p1_dot_cell_masked(a, b, level, mask, result_dev, workspace_dev, stream=0):
    check(lib().hyteg_hip_p1_dot_cell_masked(a, b, level, mask, result_dev, workspace_dev, stream), "p1_dot_cell_masked")


def p1_restrict_cell_masked(coarse, fine, coarse_level, nnc, mask, stream=0):
    check(lib().hyteg_hip_p1_restrict_cell_masked(coarse, fine, coarse_level, _f14(nnc), mask, stream),
          "p1_restrict_cell_masked")


def p1_prolongate_cell_masked(coarse, fine, coarse_level, nnc, mask, stream=0):
    check(lib().hyteg_hip_p1_prolongate_cell_masked(coarse, fine, coarse_level, _f14(nnc), mask, stream),
          "p1_prolongate_cell_masked")


def sum_shared(bases, group_ptr, entry_buf, entry_off, ngroups, n_writable, stream=0):
    check(lib().hyteg_hip_sum_shared(bases, group_ptr, entry_buf, entry_off, ngroups, n_writable, stream), "sum_shared")


def gather_entries(out, bases, entry_buf, entry_off, n, stream=0):
    check(lib().hyteg_hip_gather_entries(out, bases, entry_buf, entry_off, n, stream), "gather_entries")


def p1_copy_face_to_cell(cell, face, level, v, stream=0):
    check(lib().hyteg_hip_p1_copy_face_to_cell(cell, face, level, int(v[0]), int(v[1]), int(v[2]), stream), "p1_copy_face_to_cell")


def p1_copy_cell_to_face(face, cell, level, v, neighbor, stream=0):
    check(lib().hyteg_hip_p1_copy_cell_to_face(face, cell, level, int(v[0]), int(v[1]), int(v[2]), int(neighbor), stream),
          "p1_copy_cell_to_face")


def _ptrs(ps):
    return (C.c_void_p * len(ps))(*[int(p) for p in ps])


def _masks(ms):
    return (C.c_uint * len(ms))(*[int(m) for m in ms])


def p1_vector_cells(op, dst, srcs, scalars, level, masks, stream=0):
    """dst: list of device pointers (one per cell); srcs: list (per function) of lists (per cell)"""
    n = len(dst)
    flat = [p for f in srcs for p in f]
    sc = (C.c_double * max(1, len(scalars or [])))(*[float(v) for v in (scalars or [0.0])])
    check(lib().hyteg_hip_p1_vector_cells(op, n, _ptrs(dst), len(srcs), _ptrs(flat) if flat else None, sc if scalars is not None else None,
                                          level, _masks(masks), stream), "p1_vector_cells")


def p1_dot_cells(a, b, level, masks, result_dev, workspace_dev, stream=0):
    check(lib().hyteg_hip_p1_dot_cells(len(a), _ptrs(a), _ptrs(b), level, _masks(masks), result_dev, workspace_dev, stream), "p1_dot_cells")


def p1_apply_cells(dst, src, level, stencils_dev, masks, update=REPLACE, stream=0):
    check(lib().hyteg_hip_p1_apply_cells(len(dst), _ptrs(dst), _ptrs(src), level, stencils_dev, _masks(masks), update, stream), "p1_apply_cells")


def p1_jacobi_cells(dst, rhs, src, invdiag, level, stencils_dev, relax, masks, phase, stream=0):
    check(lib().hyteg_hip_p1_jacobi_cells(len(dst), _ptrs(dst), _ptrs(rhs), _ptrs(src), _ptrs(invdiag), level, stencils_dev, float(relax),
                                          _masks(masks), phase, stream), "p1_jacobi_cells")


def p1_restrict_cells(coarse, fine, coarse_level, nnc_inv_dev, masks, stream=0):
    check(lib().hyteg_hip_p1_restrict_cells(len(coarse), _ptrs(coarse), _ptrs(fine), coarse_level, nnc_inv_dev, _masks(masks), stream),
          "p1_restrict_cells")


def p1_prolongate_cells(coarse, fine, coarse_level, nnc_inv_dev, masks, update=REPLACE, stream=0):
    check(lib().hyteg_hip_p1_prolongate_cells(len(coarse), _ptrs(coarse), _ptrs(fine), coarse_level, nnc_inv_dev, _masks(masks), update, stream),
          "p1_prolongate_cells")


def p2_edge_array_size(level):
    return int(lib().hyteg_hip_p2_edge_array_size(level))


def p2_build_operator_table(elmat):
    """host: element matrices [6][10][10] -> operator table (numpy array) for p2_elementwise_apply_cell"""
    import numpy as np

    em = np.ascontiguousarray(elmat, dtype=np.float64).reshape(600)
    out = np.empty(int(lib().hyteg_hip_p2_operator_table_size()))
    check(lib().hyteg_hip_p2_build_operator_table(em.ctypes.data_as(_dp), out.ctypes.data_as(_dp)), "p2_build_operator_table")
    return out


def p2_edge_vector_cell_masked(op, dst, srcs, scalars, level, mask, stream=0):
    n = len(srcs)
    sc = (C.c_double * max(1, len(scalars or [])))(*[float(v) for v in (scalars or [0.0])])
    check(lib().hyteg_hip_p2_edge_vector_cell_masked(op, dst, n, _ptrs(srcs) if n else None, sc if scalars is not None else None, level, mask,
                                                     stream), "p2_edge_vector_cell_masked")


def p2_edge_dot_cell_masked(a, b, level, mask, result_dev, workspace_dev, stream=0):
    check(lib().hyteg_hip_p2_edge_dot_cell_masked(a, b, level, mask, result_dev, workspace_dev, stream), "p2_edge_dot_cell_masked")


def p2_elementwise_apply_cells(dst_v, dst_e, src_v, src_e, level, optables_dev, masks, alpha=1.0, update=REPLACE, kinds=0xFF, stream=0):
    """hyteg_hip_p2_elementwise_apply_cells_kinds: the cells of one launch (lists of device pointers, one operator table and one
    point mask per cell), levels 2..6"""
    n = len(dst_v)
    arr = lambda xs: (C.c_void_p * n)(*[int(x) for x in xs])
    check(lib().hyteg_hip_p2_elementwise_apply_cells_kinds(n, arr(dst_v), arr(dst_e), arr(src_v), arr(src_e), int(level), arr(optables_dev),
                                                           float(alpha), int(update), (C.c_uint * n)(*[int(m) for m in masks]), int(kinds),
                                                           stream), "p2_elementwise_apply_cells_kinds")


def p2_set_class_rows_min_level(level) -> int:
    """first level hyteg_hip_p2_elementwise_apply_cell uses the row kernel with every point class at (default 3; 99 = the kernels of rounds 1-2); returns the previous value"""
    return int(lib().hyteg_hip_p2_set_class_rows_min_level(int(level)))


def p2_elementwise_apply_cell(dst_v, dst_e, src_v, src_e, level, optable_dev, alpha=1.0, update=REPLACE, mask=0x7FFF, stream=0, kinds=0xFF):
    """kinds: destination kinds to compute (bit 0 vertex DoFs, 1..7 edge DoFs X, Y, Z, XY, XZ, YZ, XYZ)"""
    check(lib().hyteg_hip_p2_elementwise_apply_cell_kinds(dst_v, dst_e, src_v, src_e, level, optable_dev, float(alpha), update, mask, kinds,
                                                          stream), "p2_elementwise_apply_cell")


def p1_sor_cells(u, rhs, level, stencils_dev, relax, masks, backwards=False, stream=0):
    check(lib().hyteg_hip_p1_sor_cells(len(u), _ptrs(u), _ptrs(rhs), level, stencils_dev, float(relax), 1 if backwards else 0, _masks(masks),
                                       stream), "p1_sor_cells")


def p1_sor_shell_cells(dst, rhs, rest, level, tables_dev, relax, masks, backwards=False, stream=0):
    check(lib().hyteg_hip_p1_sor_shell_cells(len(dst), _ptrs(dst), _ptrs(rhs), _ptrs(rest), level, tables_dev, float(relax), _masks(masks),
                                             1 if backwards else 0, stream), "p1_sor_shell_cells")


def sor_shell_tables_bytes(tables):
    """pack hostutil.sor_tables()-style dicts into the byte layout of hyteg_hip_sor_shell_tables (for upload)"""
    import struct

    out = b""
    for t in tables:
        out += struct.pack("12i", *[int(v) for row in t["edge_verts"] for v in row])
        out += struct.pack("12i", *[int(v) for row in t["face_verts"] for v in row])
        out += struct.pack("18d", *[float(v) for row in t["edge_w"] for v in row])
        out += struct.pack("28d", *[float(v) for row in t["face_w"] for v in row])
        out += struct.pack("4d", *[float(v) for v in t["vertex_w"]])
    return out


def p1_sor_shell_cell(dst, rhs, rest, level, edge_verts, edge_w, face_verts, face_w, vertex_w, relax, mask, backwards=False, stream=0):
    ev = [int(v) for row in edge_verts for v in row]
    fv = [int(v) for row in face_verts for v in row]
    ew = [float(v) for row in edge_w for v in row]
    fw = [float(v) for row in face_w for v in row]
    vw = [float(v) for v in vertex_w]
    check(lib().hyteg_hip_p1_sor_shell_cell(dst, rhs, rest, level, (C.c_int * 12)(*ev), (C.c_double * 18)(*ew), (C.c_int * 12)(*fv),
                                            (C.c_double * 28)(*fw), (C.c_double * 4)(*vw), float(relax), mask, 1 if backwards else 0,
                                            stream), "p1_sor_shell_cell")


def p1_sor_face3d_workspace(level):
    return int(lib().hyteg_hip_p1_sor_face3d_workspace(level))


def p1_sor_face3d(dst, rhs, work, level, vmaps, ws, relax, backwards=False, stream=0):
    """SOR sweep over one macro-face in HyTeG's face layout (sor_3D_macroface_P1*); work: p1_sor_face3d_workspace(level) bytes."""
    n = len(vmaps)
    flat_v = [int(v) for m in vmaps for v in m]
    flat_w = [float(x) for w in ws for x in w]
    check(lib().hyteg_hip_p1_sor_face3d(dst, rhs, work, level, n, (C.c_int * len(flat_v))(*flat_v), (C.c_double * len(flat_w))(*flat_w),
                                        float(relax), int(bool(backwards)), stream), "p1_sor_face3d")


def p1_apply_face3d(dst, src, level, vmaps, ws, update=REPLACE, stream=0):
    flat_v = [int(x) for row in vmaps for x in row]
    flat_w = [float(x) for row in ws for x in row]
    n = len(flat_v) // 3
    check(lib().hyteg_hip_p1_apply_face3d(dst, src, level, n, (C.c_int * len(flat_v))(*flat_v), (C.c_double * len(flat_w))(*flat_w),
                                          update, stream), "p1_apply_face3d")


# ---- the hyteg_operators seam (generated elementwise operators: apply_macro_3D) ----
def _c12(coords):
    flat = [float(x) for v in coords for x in (v if hasattr(v, "__len__") else [v])]
    if len(flat) != 12:
        raise HytegHipError("macro_vertex_coords: 4 vertices x 3 components expected")
    return flat


def p1_elementwise_diffusion_apply_macro_3d(dst, src, coords, micro_edges, stream=0):
    a = (C.c_double * 12)(*_c12(coords))
    check(lib().hyteg_hip_p1_elementwise_diffusion_apply_macro_3d(dst, src, a, int(micro_edges), float(micro_edges), stream),
          "p1_elementwise_diffusion_apply_macro_3d")


def p1_elementwise_diffusion_apply_macro_3d_masked(dst, src, coords, micro_edges, mask=MASK_ALL, update=ADD, stream=0):
    a = (C.c_double * 12)(*_c12(coords))
    check(lib().hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_masked(dst, src, a, int(micro_edges), mask, update, stream),
          "p1_elementwise_diffusion_apply_macro_3d_masked")


def p1_elementwise_diffusion_apply_macro_3d_f32(dst, src, coords, micro_edges, stream=0):
    a = (C.c_float * 12)(*_c12(coords))
    check(lib().hyteg_hip_p1_elementwise_diffusion_apply_macro_3d_f32(dst, src, a, int(micro_edges), float(micro_edges), stream),
          "p1_elementwise_diffusion_apply_macro_3d_f32")


def p1_elementwise_diffusion_diagonal_macro_3d(diag, coords, micro_edges, stream=0):
    a = (C.c_double * 12)(*_c12(coords))
    check(lib().hyteg_hip_p1_elementwise_diffusion_diagonal_macro_3d(diag, a, int(micro_edges), float(micro_edges), stream),
          "p1_elementwise_diffusion_diagonal_macro_3d")


def p1_elementwise_diffusion_stencils(coords, micro_edges):
    """(w_inner[15], w_slots[14][15]) as Python lists: host-only, no GPU needed"""
    a = (C.c_double * 12)(*_c12(coords))
    wi, ws = (C.c_double * 15)(), (C.c_double * 210)()
    check(lib().hyteg_hip_p1_elementwise_diffusion_stencils(a, int(micro_edges), wi, ws), "p1_elementwise_diffusion_stencils")
    return list(wi), [list(ws[15 * s:15 * s + 15]) for s in range(14)]


def p1_apply_cell_boundary_f32(dst, src, level, w_slots, mask=MASK_SHELL, update=REPLACE, stream=0):
    flat = [float(x) for row in w_slots for x in row]
    a = (C.c_double * 210)(*flat)
    check(lib().hyteg_hip_p1_apply_cell_boundary_f32(dst, src, level, a, mask, update, stream), "p1_apply_cell_boundary_f32")


def p2_elementwise_diffusion_apply_macro_3d(dst_v, dst_e, src_v, src_e, coords, micro_edges, stream=0):
    a = (C.c_double * 12)(*_c12(coords))
    check(lib().hyteg_hip_p2_elementwise_diffusion_apply_macro_3d(dst_v, dst_e, src_v, src_e, a, int(micro_edges), float(micro_edges), stream),
          "p2_elementwise_diffusion_apply_macro_3d")


def p2_elementwise_diffusion_element_matrices(coords, micro_edges):
    a = (C.c_double * 12)(*_c12(coords))
    out = (C.c_double * 600)()
    check(lib().hyteg_hip_p2_elementwise_diffusion_element_matrices(a, int(micro_edges), out), "p2_elementwise_diffusion_element_matrices")
    return list(out)


# ---- the constant-stencil P2 operator's kernel seam (P2ConstantOperator's four sub-operators) ----
def p2_constant_stencil_layout():
    """(counts [v2v, e2v, v2e, e2e], keys [(dst kind, src kind, dx, dy, dz), ...] in the order a binding flattens its maps)"""
    counts = (C.c_int * 4)()
    check(lib().hyteg_hip_p2_constant_stencil_layout(counts, None), "p2_constant_stencil_layout")
    total = sum(counts)
    keys = (C.c_int * (5 * total))()
    check(lib().hyteg_hip_p2_constant_stencil_layout(counts, keys), "p2_constant_stencil_layout")
    return list(counts), [tuple(keys[5 * i:5 * i + 5]) for i in range(total)]


def p2_build_operator_table_from_stencils(inner, classes=None):
    n = int(lib().hyteg_hip_p2_operator_table_size())
    a = (C.c_double * len(inner))(*[float(x) for x in inner])
    b = None if classes is None else (C.c_double * len(classes))(*[float(x) for x in classes])
    out = (C.c_double * n)()
    check(lib().hyteg_hip_p2_build_operator_table_from_stencils(a, b, out), "p2_build_operator_table_from_stencils")
    return list(out)


def p2_operator_table_closure_split(table):
    """-> (outside, closure_vertex, closure_edge): the boundary-class rows of an operator table split by where the source lies"""
    n = int(lib().hyteg_hip_p2_operator_table_size())
    t = (C.c_double * n)(*[float(x) for x in table])
    o, v, e = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    check(lib().hyteg_hip_p2_operator_table_closure_split(t, o, v, e), "p2_operator_table_closure_split")
    return list(o), list(v), list(e)


def p2_operator_table_face_edge_weights(table, face_verts):
    """-> [3][5]: this cell's share of the couplings between the edge DoFs inside a macro-face, in the face's frame"""
    n = int(lib().hyteg_hip_p2_operator_table_size())
    t = (C.c_double * n)(*[float(x) for x in table])
    fv = (C.c_int * 3)(*[int(x) for x in face_verts])
    w = (C.c_double * 15)()
    check(lib().hyteg_hip_p2_operator_table_face_edge_weights(t, fv, w), "p2_operator_table_face_edge_weights")
    return [list(w[5 * k:5 * k + 5]) for k in range(3)]


def p2_sor_face_edgedofs_cell(dst_edge, q_edge, level, face_verts, face_w, relax, mask, backwards=False, stream=0):
    fv = (C.c_int * 12)(*[int(x) for f in face_verts for x in f])
    fw = (C.c_double * 60)(*[float(x) for f in face_w for t in f for x in t])
    check(lib().hyteg_hip_p2_sor_face_edgedofs_cell(dst_edge, q_edge, level, fv, fw, float(relax), mask, 1 if backwards else 0, stream),
          "p2_sor_face_edgedofs_cell")


def _edge_blocks(base, level):
    """the seven block pointers in the reference kernels' (alphabetical) order X, XY, XYZ, XZ, Y, YZ, Z"""
    n = 1 << level
    b = 8 * (n * (n + 1) * (n + 2) // 6)
    x, y, z, xy, xz, yz, xyz = (base + k * b for k in range(7))
    return [x, xy, xyz, xz, y, yz, z]


def p2_apply_cell_edgedof_to_vertexdof(src_edge, dst_vertex, e2v, level, update=REPLACE, stream=0):
    w = (C.c_double * len(e2v))(*[float(x) for x in e2v])
    check(lib().hyteg_hip_p2_apply_cell_edgedof_to_vertexdof(*_edge_blocks(src_edge, level), dst_vertex, w, level, update, stream),
          "p2_apply_cell_edgedof_to_vertexdof")


def p2_apply_cell_vertexdof_to_edgedof(dst_edge, src_vertex, v2e, level, update=REPLACE, stream=0):
    w = (C.c_double * len(v2e))(*[float(x) for x in v2e])
    check(lib().hyteg_hip_p2_apply_cell_vertexdof_to_edgedof(*_edge_blocks(dst_edge, level), src_vertex, level, w, update, stream),
          "p2_apply_cell_vertexdof_to_edgedof")


def p2_apply_cell_edgedof_to_edgedof(dst_edge, src_edge, e2e, level, update=REPLACE, stream=0):
    w = (C.c_double * len(e2e))(*[float(x) for x in e2e])
    check(lib().hyteg_hip_p2_apply_cell_edgedof_to_edgedof(*_edge_blocks(dst_edge, level), *_edge_blocks(src_edge, level), w, level, update, stream),
          "p2_apply_cell_edgedof_to_edgedof")
