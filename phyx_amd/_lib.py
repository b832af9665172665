"""ctypes loader for libphyx_amd.so — the C-ABI boundary (include/phyx_amd.h).

The library is the product; there is no Python or CPU fallback.  Loading fails loudly when the
shared object is missing, and every compute call fails with PHX_ERR_NO_DEVICE when there is no GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libphyx_amd.so")

PHX_OK, PHX_ERR_INVALID, PHX_ERR_NO_DEVICE, PHX_ERR_HIP, PHX_ERR_CAPACITY, PHX_ERR_STATE = 0, -1, -2, -3, -4, -5


class PhxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libphyx_amd: status %d: %s" % (status, message))
        self.status = status


class Config(C.Structure):
    """phx_config == Configuration (ref: src/Configuration.h:20-23)."""
    _fields_ = [("solve_mode", C.c_int32), ("island_mode", C.c_int32),
                ("contact_iterations", C.c_int32), ("penetration_iterations", C.c_int32)]


class BodyView(C.Structure):
    """phx_body_view: the resident structure-of-arrays form of body state (three device float4 arrays)."""
    _fields_ = [("vel", C.c_void_p), ("dvel", C.c_void_p), ("mpos", C.c_void_p)]


class SolveStats(C.Structure):
    _fields_ = [("island_count", C.c_int32), ("island_max_size", C.c_int32), ("colour_count", C.c_int32),
                ("impulse_iterations", C.c_int32), ("displacement_iterations", C.c_int32),
                ("lds_islands", C.c_int32), ("recoloured", C.c_int32), ("graph_replay", C.c_int32),
                ("device_ms", C.c_double), ("joint_visits", C.c_int64)]


class BroadphaseStats(C.Structure):
    _fields_ = [("candidate_tests", C.c_int64), ("overlapping_pairs", C.c_int64), ("new_pairs", C.c_int32),
                ("set_size", C.c_int32), ("device_ms", C.c_double)]


class BenchResult(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("impulse_kernel_ms", C.c_double), ("impulse_launches", C.c_int64),
                ("joint_visits", C.c_int64), ("impulse_iterations", C.c_int64), ("bracketed_launches", C.c_int64)]


SLAB_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)      # (user, send, recv, bytes_per_rank), host buffers
SLAB_ALL_REDUCE_MAX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64))              # (user, value in place)


class SlabTransport(C.Structure):
    """phx_slab_transport: how the ranks of a re-slab talk (include/phyx_amd.h)."""
    _fields_ = [("rank", C.c_int32), ("size", C.c_int32), ("comm", C.c_void_p), ("all_gather", SLAB_ALL_GATHER), ("all_reduce_max", SLAB_ALL_REDUCE_MAX),
                ("user", C.c_void_p)]


_lib = None

_vp, _i32, _f32 = C.c_void_p, C.c_int32, C.c_float
STEP_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32)      # phx_step_hook(user, step, phase)
_SIGNATURES = {
    "phx_abi_version": (C.c_int, []),
    "phx_arith_mode": (C.c_int, []),
    "phx_last_error": (C.c_char_p, []),
    "phx_device_count": (C.c_int, []),
    "phx_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "phx_device_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_vp)]),
    "phx_device_free": (C.c_int, [C.c_int, _vp]),
    "phx_memcpy_h2d": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t]),
    "phx_memcpy_d2h": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t]),
    "phx_memcpy_d2d": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t]),
    "phx_memcpy_d2h_on": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "phx_memcpy_h2d_on": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "phx_memcpy_d2d_on": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "phx_debug_wait_clock": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "phx_exchange_layout": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, C.POINTER(C.c_int64)]),
    "phx_solver_set_exchange_buffers": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "phx_solver_exchange_pack": (C.c_int, [_vp, _vp, _vp, _i32, C.POINTER(C.c_size_t)]),
    "phx_solver_exchange_unpack": (C.c_int, [_vp, _vp, _vp]),
    "phx_solver_exchange_status": (C.c_int, [_vp, C.POINTER(_i32)]),
    "phx_solver_exchange_segment_bytes": (C.c_size_t, [_vp]),
    "phx_comm_unique_id": (C.c_int, [_vp]),
    "phx_comm_create": (C.c_int, [C.POINTER(_vp), _vp, _i32, _i32, C.c_int]),
    "phx_comm_destroy": (None, [_vp]),
    "phx_comm_rccl_version": (C.c_int, []),
    "phx_comm_rank": (C.c_int, [_vp]),
    "phx_comm_size": (C.c_int, [_vp]),
    "phx_comm_all_gather": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "phx_comm_barrier": (C.c_int, [_vp, _vp]),
    "phx_comm_barrier_async": (C.c_int, [_vp, _vp]),
    "phx_comm_async_error": (C.c_int, [_vp, C.POINTER(_i32)]),
    "phx_solver_set_comm": (C.c_int, [_vp, _vp]),
    "phx_world_set_comm": (C.c_int, [_vp, _vp]),
    "phx_world_step_sharded": (C.c_int, [_vp, _f32, C.POINTER(Config)]),
    "phx_world_check_exchange": (C.c_int, [_vp]),
    "phx_solver_solve_resident": (C.c_int, [_vp, C.POINTER(BodyView), _i32, _vp, _i32, _vp, _i32, C.POINTER(Config)]),
    "phx_bodies_to_view": (C.c_int, [C.c_int, _vp, _i32, C.POINTER(BodyView), _vp]),
    "phx_view_to_bodies": (C.c_int, [C.c_int, C.POINTER(BodyView), _i32, _vp, _vp]),
    "phx_world_step_begin": (C.c_int, [_vp, _f32, C.POINTER(Config), C.POINTER(C.c_size_t)]),
    "phx_world_step_end": (C.c_int, [_vp, _f32]),
    "phx_world_stream": (C.c_void_p, [_vp]),
    "phx_solver_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "phx_solver_destroy": (None, [_vp]),
    "phx_solver_solve": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, C.POINTER(Config)]),
    "phx_solver_solve_device": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, C.POINTER(Config)]),
    "phx_solver_synchronize": (C.c_int, [_vp]),
    "phx_solver_set_schedule_reuse": (C.c_int, [_vp, _i32]),
    "phx_solver_set_trace": (C.c_int, [_vp, _i32]),
    "phx_solver_get_wave_trace": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i32)]),
    "phx_solver_get_island_trace": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i32)]),
    "phx_solver_set_shard": (C.c_int, [_vp, _i32, _i32]),
    "phx_solver_set_body_state_bits": (C.c_int, [_vp, _i32]),
    "phx_solver_get_stats": (C.c_int, [_vp, C.POINTER(SolveStats)]),
    "phx_solver_get_schedule": (C.c_int, [_vp, _vp, _i32, _vp, _i32, C.POINTER(_i32)]),
    "phx_solver_get_groups": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "phx_solver_get_partition": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "phx_solver_get_lanes": (C.c_int, [_vp, _vp, _vp, _i32, C.POINTER(_i32)]),
    "phx_solver_get_refreshed": (C.c_int, [_vp, _i32, _vp]),
    "phx_solver_bench": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, C.POINTER(Config), _i32, _i32, C.POINTER(BenchResult)]),
    "phx_solver_bench_stage": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32]),
    "phx_solver_bench_checksum": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "phx_solver_bench_hooked": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, C.POINTER(Config), _i32, _i32, STEP_HOOK, _vp, C.POINTER(BenchResult)]),
    "phx_solver_stream": (C.c_void_p, [_vp]),
    "phx_schedule_priority": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "phx_schedule_colours": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, C.POINTER(_i32)]),
    "phx_schedule_islands": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _i32]),
    "phx_schedule_groups": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "phx_broadphase_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "phx_broadphase_destroy": (None, [_vp]),
    "phx_broadphase_clear": (C.c_int, [_vp]),
    "phx_broadphase_update": (C.c_int, [_vp, _vp, _i32, _vp, _i32, C.POINTER(_i32)]),
    "phx_broadphase_update_device": (C.c_int, [_vp, _vp, _i32]),
    "phx_broadphase_get_sorted": (C.c_int, [_vp, _vp, _vp, _i32]),
    "phx_broadphase_get_new_pairs": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i32)]),
    "phx_broadphase_erase_pairs": (C.c_int, [_vp, _vp, _i32]),
    "phx_broadphase_get_stats": (C.c_int, [_vp, C.POINTER(BroadphaseStats)]),
    "phx_world_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "phx_world_destroy": (None, [_vp]),
    "phx_world_add_body": (C.c_int, [_vp, _f32, _f32, _f32, _f32, _f32]),
    "phx_world_set_body_static": (C.c_int, [_vp, _i32]),
    "phx_world_set_body_inverse_mass": (C.c_int, [_vp, _i32, _f32, _f32]),
    "phx_world_set_gravity": (C.c_int, [_vp, _f32]),
    "phx_world_set_shard": (C.c_int, [_vp, _i32, _i32]),
    "phx_world_update": (C.c_int, [_vp, _f32, C.POINTER(Config)]),
    "phx_world_pre_solve": (C.c_int, [_vp, _f32]),
    "phx_world_finish_step": (C.c_int, [_vp, _f32, C.POINTER(Config)]),
    "phx_world_counts": (C.c_int, [_vp] + [C.POINTER(_i32)] * 4),
    "phx_world_get_bodies": (C.c_int, [_vp, _vp, _i32]),
    "phx_world_get_manifolds": (C.c_int, [_vp, _vp, _i32]),
    "phx_world_get_contact_points": (C.c_int, [_vp, _vp, _i32]),
    "phx_world_get_joints": (C.c_int, [_vp, _vp, _i32]),
    "phx_world_set_state": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32]),
    "phx_world_get_solve_stats": (C.c_int, [_vp, C.POINTER(SolveStats)]),
    "phx_world_get_broadphase_stats": (C.c_int, [_vp, C.POINTER(BroadphaseStats)]),
    "phx_world_solver": (_vp, [_vp]),
    "phx_world_broadphase": (_vp, [_vp]),
    "phx_world_get_phase_ms": (C.c_int, [_vp, _vp]),
    "phx_world_set_phase_timing": (C.c_int, [_vp, _i32]),
    "phx_world_debug_counters": (C.c_int, [_vp, _vp]),
    "phx_world_build_counts": (C.c_int, [_vp, _vp]),
    "phx_world_x_extent": (C.c_int, [_vp, _vp]),
    "phx_world_reslab": (C.c_int, [_vp, C.POINTER(SlabTransport), _vp, _i32, C.POINTER(_i32), _i32, C.c_double, _vp, C.POINTER(_i32)]),
    "phx_world_reslab_intervals": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, C.POINTER(_i32)]),
    "phx_reslab_plan": (C.c_int, [_vp, _vp, _vp, _i32, _i32, C.c_double, _vp, _vp]),
    "phx_reslab_cuts": (C.c_int, [_vp, _vp, _i32, _i32, C.c_double, _vp, _vp]),
    "phx_world_synchronize": (C.c_int, [_vp]),
}


def declared_symbols():
    return sorted(_SIGNATURES)


def load():
    """Load the shared library (build it first with phyx_amd.build or __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libphyx_amd.so is not built: run `python -m phyx_amd.build` (needs hipcc). "
                              "There is no fallback implementation.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError here = header/library drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status):
    if status < 0:
        raise PhxError(status, (load().phx_last_error() or b"").decode("utf-8", "replace"))
    return status
