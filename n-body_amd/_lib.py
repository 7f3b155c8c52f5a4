"""ctypes binding of libnbody_hip.so (the C ABI declared in include/nbody_hip.h).

There is deliberately no fallback: if the shared library is missing, or no HIP device is
present when a context is created, the call raises.  Nothing here imports `oracle/`.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import sys
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
# NBODY_HIP_LIB: another build of the same library (kernel experiments: tools/ build variants into gpurun_out/)
LIB_PATH = os.environ.get("NBODY_HIP_LIB") or os.path.join(_HERE, "lib", "libnbody_hip.so")

# status codes (include/nbody_hip.h: nbody_hip_status)
OK, ERR_VALIDATION, ERR_DEVICE, ERR_RESOURCE, ERR_STATE, ERR_COMM = 0, -1, -2, -3, -4, -5


class NBodyError(RuntimeError):
    """Base of the errors raised from the C ABI (ref: include/nbody/error_handling.hpp:29-102)."""


class ValidationException(NBodyError, ValueError):
    """ref: nbody::ValidationException"""


class DeviceException(NBodyError):
    """ref: nbody::CudaException (HIP runtime error here)"""


class ResourceException(NBodyError, MemoryError):
    """ref: nbody::ResourceException"""


class StateException(NBodyError):
    """call-sequence error (null handle / missing arrays)"""


class CommException(NBodyError):
    """RCCL error / RCCL not loadable (include/nbody_hip_comm.h)"""


_EXC = {ERR_VALIDATION: ValidationException, ERR_DEVICE: DeviceException,
        ERR_RESOURCE: ResourceException, ERR_STATE: StateException, ERR_COMM: CommException}


class ParticleDataStruct(C.Structure):
    """Layout-identical to nbody::ParticleData (include/nbody/types.hpp:234-276)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "acc_x", "acc_y", "acc_z",
        "acc_old_x", "acc_old_y", "acc_old_z", "mass")] + [("count", C.c_size_t)]


class DirectInfoStruct(C.Structure):
    """nbody_hip_direct_info_t (include/nbody_hip.h)"""
    _fields_ = [("deterministic_mode", C.c_int), ("kernel", C.c_int), ("bodies_per_lane_equal", C.c_int),
                ("bodies_per_lane_general", C.c_int), ("reaction_slots", C.c_int), ("iside_slots", C.c_int),
                ("reserved0", C.c_int), ("reserved1", C.c_int), ("workspace_bytes_needed", C.c_ulonglong),
                ("slot_bytes_wanted", C.c_ulonglong), ("workspace_bytes_held", C.c_ulonglong),
                ("last_kernel", C.c_int), ("reserved2", C.c_int)]


FIELDS = tuple(n for n, _ in ParticleDataStruct._fields_[:13])

# every symbol include/nbody_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_PD = C.POINTER(ParticleDataStruct)
PROTOTYPES = {
    "nbody_hip_abi_version": (C.c_int, []),
    "nbody_hip_sort_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nbody_hip_last_error": (C.c_char_p, []),
    "nbody_hip_device_count": (C.c_int, []),
    "nbody_hip_ctx_create": (C.c_int, [C.POINTER(_P), C.c_int, _P]),
    "nbody_hip_ctx_destroy": (C.c_int, [_P]),
    "nbody_hip_ctx_set_stream": (C.c_int, [_P, _P]),
    "nbody_hip_ctx_synchronize": (C.c_int, [_P]),
    "nbody_hip_capture_begin": (C.c_int, [_P]),
    "nbody_hip_capture_end": (C.c_int, [_P, C.POINTER(_P)]),
    "nbody_hip_graph_launch": (C.c_int, [_P, C.c_int]),
    "nbody_hip_graph_destroy": (C.c_int, [_P]),
    "nbody_hip_particles_alloc": (C.c_int, [_PD, C.c_size_t]),
    "nbody_hip_particles_free": (C.c_int, [_PD]),
    "nbody_hip_particles_upload": (C.c_int, [_PD, _PD]),
    "nbody_hip_particles_download": (C.c_int, [_PD, _PD]),
    "nbody_hip_direct_forces": (C.c_int, [_P, _PD, C.c_float, C.c_float, C.c_int]),
    "nbody_hip_direct_forces_packed": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, _P,
                                                 C.c_float, C.c_float, C.c_int]),
    "nbody_hip_direct_forces_pair_packed": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, _P, C.c_int, _P,
                                                      C.c_int, C.c_float, C.c_float]),
    "nbody_hip_pack_posm": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, _P]),
    "nbody_hip_unpack3": (C.c_int, [_P, _P, C.c_size_t, _P, _P, _P]),
    "nbody_hip_update_positions": (C.c_int, [_P, _PD, C.c_float]),
    "nbody_hip_update_velocities": (C.c_int, [_P, _PD, C.c_float]),
    "nbody_hip_store_accelerations": (C.c_int, [_P, _PD]),
    "nbody_hip_drift": (C.c_int, [_P, _PD, C.c_float]),
    "nbody_hip_integrate_direct": (C.c_int, [_P, _PD, C.c_float, C.c_float, C.c_float, C.c_int]),
    "nbody_hip_drift_packed": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.c_float]),
    "nbody_hip_kick_packed": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.c_float]),
    "nbody_hip_kinetic_energy": (C.c_int, [_P, _PD, C.POINTER(C.c_float)]),
    "nbody_hip_potential_energy": (C.c_int, [_P, _PD, C.c_float, C.c_float, C.POINTER(C.c_float)]),
    "nbody_hip_kinetic_energy_f64": (C.c_int, [_P, _PD, C.POINTER(C.c_double)]),
    "nbody_hip_potential_energy_f64": (C.c_int, [_P, _PD, C.c_float, C.c_float,
                                                 C.POINTER(C.c_double)]),
    "nbody_hip_energies_packed": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_longlong, _P, C.c_size_t, C.c_float, C.c_float,
                                            C.POINTER(C.c_double)]),
    "nbody_hip_grid_create": (C.c_int, [_P, C.c_size_t, C.c_float, C.POINTER(_P)]),
    "nbody_hip_grid_destroy": (C.c_int, [_P]),
    "nbody_hip_grid_set_cell_size": (C.c_int, [_P, C.c_float]),
    "nbody_hip_grid_tuning": (C.c_int, [_P, C.c_int]),
    "nbody_hip_grid_build": (C.c_int, [_P, _PD]),
    "nbody_hip_grid_drift_build": (C.c_int, [_P, _PD, C.c_float]),
    "nbody_hip_grid_compute_forces": (C.c_int, [_P, _PD, C.c_float, C.c_float, C.c_float]),
    "nbody_hip_grid_info": (C.c_int, [_P, C.POINTER(C.c_int * 3), C.POINTER(C.c_int),
                                      C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3)]),
    "nbody_hip_grid_count": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "nbody_hip_grid_copy_cell_data": (C.c_int, [_P, _P, _P, _P, _P]),
    "nbody_hip_grid_build_packed": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "nbody_hip_grid_compute_forces_packed": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, _P]),
    "nbody_hip_grid_sorted_bodies": (C.c_int, [_P, C.c_size_t, C.c_size_t, _P]),
    "nbody_hip_grid_set_slab": (C.c_int, [_P, C.c_int, C.c_int]),
    "nbody_hip_grid_forces_pair_packed": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _P,
                                                    C.c_int]),
    "nbody_hip_grid_export_layer": (C.c_int, [_P, C.c_int, _P, _P]),
    "nbody_hip_grid_forces_layer_packed": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_float, C.c_float, C.c_float, _P, C.c_int]),
    "nbody_hip_slab_partition": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, _P, C.c_float, C.c_int, C.c_int, C.c_int,
                                           _P, _P, _P, _P, _P]),
    "nbody_hip_slab_partition_cuts": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, _P, C.c_float, C.c_int, C.c_int, C.c_int,
                                                _P, _P, _P, _P, _P, _P]),
    "nbody_hip_slab_layer_owner": (C.c_int, [C.c_int, C.c_float, C.c_float, C.c_int, _P]),
    "nbody_hip_slab_fill": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, C.c_size_t, _P, _P, _P, _P]),
    "nbody_hip_bbox_packed": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "nbody_hip_drift_bbox_packed": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.c_float, _P, _P]),
    "nbody_hip_cell_z_packed": (C.c_int, [_P, _P, C.c_size_t, C.c_float, C.c_float, C.c_int, _P]),
    "nbody_hip_tree_create": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "nbody_hip_tree_destroy": (C.c_int, [_P]),
    "nbody_hip_tree_set_params": (C.c_int, [_P, C.c_int, C.c_int]),
    "nbody_hip_tree_limit_nodes": (C.c_int, [_P, C.c_int]),
    "nbody_hip_tree_tuning": (C.c_int, [_P, C.c_int, C.c_int]),
    "nbody_hip_tree_walk_form": (C.c_int, [_P, C.c_int]),
    "nbody_hip_tree_visit_histogram": (C.c_int, [_P, C.POINTER(C.c_ulonglong * 130)]),
    "nbody_hip_tree_count_visits": (C.c_int, [_P, C.c_int]),
    "nbody_hip_tree_build": (C.c_int, [_P, _PD]),
    "nbody_hip_tree_drift_build": (C.c_int, [_P, _PD, C.c_float]),
    "nbody_hip_tree_compute_forces": (C.c_int, [_P, _PD, C.c_float, C.c_float, C.c_float]),
    "nbody_hip_tree_build_packed": (C.c_int, [_P, _P, C.c_size_t]),
    "nbody_hip_tree_compute_forces_packed": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_float, _P]),
    "nbody_hip_tree_stats": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_float),
                                       C.POINTER(C.c_ulonglong), C.POINTER(C.c_int * 24)]),
    "nbody_hip_tree_copy_nodes": (C.c_int, [_P, _P, C.c_int, _P]),
    "nbody_hip_time_direct_packed": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, _P, C.c_float,
                                               C.c_float, C.c_int, C.POINTER(C.c_float)]),
    "nbody_hip_direct_deterministic": (C.c_int, [_P, C.c_int]),
    "nbody_hip_direct_slot_budget": (C.c_int, [_P, C.c_ulonglong]),
    "nbody_hip_direct_info": (C.c_int, [_P, C.c_size_t, C.c_float, _P]),
    "nbody_hip_direct_tuning": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    # include/nbody_hip_comm.h
    "nbody_hip_comm_init_all": (C.c_int, [C.c_int, _P, C.c_int, C.POINTER(_P)]),
    "nbody_hip_comm_unique_id": (C.c_int, [_P]),
    "nbody_hip_comm_init_rank": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "nbody_hip_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _P]),
    "nbody_hip_comm_destroy": (C.c_int, [_P]),
    "nbody_hip_shard_bounds": (C.c_int, [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_size_t)]),
    "nbody_hip_pair_schedule": (C.c_int, [C.c_int, C.c_int, C.c_size_t, _P, C.c_int]),
    "nbody_hip_sharded_direct_create": (C.c_int, [_P, C.c_size_t, C.c_float, C.c_float, C.POINTER(_P)]),
    "nbody_hip_sharded_direct_destroy": (C.c_int, [_P]),
    "nbody_hip_sharded_direct_set_state": (C.c_int, [_P] * 8),
    "nbody_hip_sharded_direct_forces": (C.c_int, [_P]),
    "nbody_hip_sharded_direct_step": (C.c_int, [_P, C.c_float, C.c_int]),
    "nbody_hip_sharded_direct_time_steps": (C.c_int, [_P, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "nbody_hip_sharded_direct_synchronize": (C.c_int, [_P]),
    "nbody_hip_sharded_direct_get_state": (C.c_int, [_P] * 10 + [C.c_int]),
    "nbody_hip_sharded_direct_energies": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nbody_hip_sharded_direct_compute_forces": (C.c_int, [_P, _PD]),
    "nbody_hip_sharded_hash_create": (C.c_int, [_P, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(_P)]),
    "nbody_hip_sharded_hash_destroy": (C.c_int, [_P]),
    "nbody_hip_sharded_hash_set_state": (C.c_int, [_P] * 8),
    "nbody_hip_sharded_hash_forces": (C.c_int, [_P]),
    "nbody_hip_sharded_hash_step": (C.c_int, [_P, C.c_float, C.c_int]),
    "nbody_hip_sharded_hash_time_steps": (C.c_int, [_P, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "nbody_hip_sharded_hash_synchronize": (C.c_int, [_P]),
    "nbody_hip_sharded_hash_get_state": (C.c_int, [_P] * 10),
    "nbody_hip_sharded_hash_info": (C.c_int, [_P, _P, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), _P]),
}

_lib = None

# ---- handle lifetime ------------------------------------------------------------------------
# Every Python object that owns a C-ABI handle registers here (weakly).  ONE atexit hook closes whatever is still
# alive, in dependency order (systems before the communicators and contexts they borrow), while the HIP runtime is
# still up; `__del__` does nothing once the interpreter is finalising.  Without this, an object kept alive by a
# traceback (a failed test under `pytest -x`) was collected during interpreter finalisation, after the HIP runtime's own
# static destructors: nbody_hip_tree_destroy -> hipStreamSynchronize on a dead runtime threw std::bad_variant_access
# inside the runtime and the process ended with SIGABRT (rc 134) instead of the test's exit code.
# ref dtor this mirrors: src/cuda/force_barnes_hut.cu:212-216 (frees in ~BarnesHutTree, while the CUDA runtime lives).
CLOSE_ORDER = ("system", "graph", "tree", "grid", "backend", "comm", "context")
_live = {kind: weakref.WeakSet() for kind in CLOSE_ORDER}
_hook_registered = False
_closing_all = False


def track(obj, kind: str):
    """Registers a handle owner (it must have close()) for the exit hook.  Returns obj."""
    global _hook_registered
    _live[kind].add(obj)
    if not _hook_registered:
        atexit.register(close_all)
        _hook_registered = True
    return obj


def finalizing() -> bool:
    """True once `__del__` must not call into the library any more."""
    return _closing_all or sys.is_finalizing()


def close_all():
    """Closes every live handle owner in dependency order (the atexit hook; also callable by hand)."""
    global _closing_all
    for kind in CLOSE_ORDER:
        for obj in list(_live[kind]):
            try:
                obj.close()
            except Exception:
                pass
        _live[kind].clear()
    _closing_all = sys.is_finalizing()


def load():
    """Loads libnbody_hip.so (once).  Raises ImportError with the build hint if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C n-body_amd/csrc` (hipcc, gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.nbody_hip_abi_version() != 1:
        raise ImportError("libnbody_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int):
    if rc == OK:
        return
    msg = load().nbody_hip_last_error().decode("utf-8", "replace")
    raise _EXC.get(rc, NBodyError)(msg)
