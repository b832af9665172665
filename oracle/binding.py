"""ctypes binding of the CPU oracle (oracle/liboracle.so) and of the reference leaf shims
(oracle/_ref/libphyx_ref_leaf.so).

TEST INFRASTRUCTURE.  May be imported only from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under phyx_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# POD layouts, byte-identical to the reference (RigidBody.h:12-57, Manifold.h, Joints.h, Collider.h:45-56)
vec2 = np.dtype([("x", "<f4"), ("y", "<f4")])
body_dtype = np.dtype([
    ("index", "<u4"), ("geom_size", vec2), ("geom_xv", vec2), ("geom_yv", vec2), ("geom_pos", vec2),
    ("aabb_min", vec2), ("aabb_max", vec2), ("velocity", vec2), ("acceleration", vec2),
    ("displacing_velocity", vec2), ("angular_velocity", "<f4"), ("angular_acceleration", "<f4"),
    ("displacing_angular_velocity", "<f4"), ("inv_mass", "<f4"), ("inv_inertia", "<f4"),
    ("xv", vec2), ("yv", vec2), ("pos", vec2), ("last_iteration", "<i4"), ("last_displacement_iteration", "<i4"),
])
contact_point_dtype = np.dtype([
    ("delta1", vec2), ("delta2", vec2), ("normal", vec2), ("is_merged", "u1"), ("is_newly_created", "u1"),
    ("pad", "u1", (2,)), ("solver_index", "<i4"),
])
manifold_dtype = np.dtype([("body1", "<i4"), ("body2", "<i4"), ("point_count", "<i4"), ("point_index", "<i4")])
joint_dtype = np.dtype([("contact_point_index", "<i4"), ("body1", "<i4"), ("body2", "<i4"),
                        ("normal_acc", "<f4"), ("friction_acc", "<f4")])
bp_entry_dtype = np.dtype([("minx", "<f4"), ("maxx", "<f4"), ("centery", "<f4"), ("extenty", "<f4"), ("index", "<u4")])
sort_entry_dtype = np.dtype([("value", "<u4"), ("index", "<u4")])
assert body_dtype.itemsize == 128 and contact_point_dtype.itemsize == 32
assert manifold_dtype.itemsize == 16 and joint_dtype.itemsize == 20 and bp_entry_dtype.itemsize == 20

SOLVE_SCALAR, SOLVE_SSE2, SOLVE_AVX2 = 0, 1, 2
ISLAND_SINGLE, ISLAND_MULTIPLE, ISLAND_SINGLE_SLOPPY, ISLAND_MULTIPLE_SLOPPY = 0, 1, 2, 3
STAG_SEQUENTIAL, STAG_COLOUR_SYNC = 0, 1


class SolveStats(C.Structure):
    _fields_ = [("island_count", C.c_int32), ("island_max_size", C.c_int32), ("group_offset", C.c_int32),
                ("impulse_iterations", C.c_int32), ("displacement_iterations", C.c_int32),
                ("joint_visits", C.c_int64), ("joints_computed", C.c_int64), ("stag_events", C.c_int64)]


def build(force=False):
    """Compile liboracle.so (always possible) and, when /root/reference exists, the leaf shims."""
    so = os.path.join(_HERE, "liboracle.so")
    src_newer = (not os.path.exists(so)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so) for f in ("phx_oracle.c", "phx_oracle.h"))
    if force or src_newer:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if force:
        subprocess.check_call(["make", "-B", "-C", _HERE, "libcpubaseline_fast.so", "libcpubaseline_strict.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libphyx_ref_leaf.so")
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(ref_so)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def _p(arr):
    return arr.ctypes.data_as(C.c_void_p)


def lib():
    global _lib
    if _lib is None:
        build()
        # PHX_ORACLE_SAN=1: the ASan + UBSan build (`make -C oracle asan`; run under LD_PRELOAD of libasan, `make -C oracle asan_test`)
        L = C.CDLL(os.path.join(_HERE, "liboracle_asan.so" if os.environ.get("PHX_ORACLE_SAN") == "1" else "liboracle.so"))
        L.phxo_radix_float.restype = C.c_uint32
        L.phxo_radix_float.argtypes = [C.c_float]
        L.phxo_pair_hash.restype = C.c_uint32
        L.phxo_pair_hash.argtypes = [C.c_uint32, C.c_uint32]
        L.phxo_radix_sort3.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.phxo_body_init.argtypes = [C.c_void_p] + [C.c_float] * 6
        L.phxo_recompute_aabb.argtypes = [C.c_void_p]
        L.phxo_rotate_vec.argtypes = [C.c_void_p, C.c_float]
        L.phxo_support_points.restype = C.c_int
        L.phxo_support_points.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.phxo_contact_equals.restype = C.c_int
        L.phxo_contact_equals.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        L.phxo_contact_point_make.argtypes = [C.c_void_p] + [C.c_float] * 6 + [C.c_void_p, C.c_void_p]
        L.phxo_project_point_to_line.argtypes = [C.c_float] * 8 + [C.c_void_p]
        L.phxo_flipsign.restype = C.c_float
        L.phxo_flipsign.argtypes = [C.c_float, C.c_float, C.c_int]
        L.phxo_set_arith.argtypes = [C.c_int]
        L.phxo_get_arith.restype = C.c_int
        L.phxo_max.restype = C.c_float
        L.phxo_max.argtypes = [C.c_float, C.c_float]
        L.phxo_aabb_intersects.restype = C.c_int
        L.phxo_aabb_intersects.argtypes = [C.c_void_p, C.c_void_p]
        L.phxo_broadphase_build.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.phxo_sweep_candidates.restype = C.c_size_t
        L.phxo_sweep_candidates.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.phxo_solver_solve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.phxo_solver_solve_ordered.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.phxo_solver_solve_grouped.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.phxo_solver_solve_grouped_fp16.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                     C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.phxo_refresh_joint.argtypes = [C.c_void_p] * 4
        L.phxo_gather_islands.restype = C.c_int
        L.phxo_gather_islands.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.phxo_prepare_indices.restype = C.c_int
        L.phxo_prepare_indices.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.phxo_world_create.restype = C.c_void_p
        L.phxo_world_destroy.argtypes = [C.c_void_p]
        L.phxo_world_add_body.restype = C.c_int
        L.phxo_world_add_body.argtypes = [C.c_void_p] + [C.c_float] * 5
        L.phxo_world_set_gravity.argtypes = [C.c_void_p, C.c_float]
        L.phxo_world_update.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.phxo_world_pre_solve.argtypes = [C.c_void_p, C.c_float]
        L.phxo_world_solve_and_integrate.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.phxo_world_integrate_position.argtypes = [C.c_void_p, C.c_float]
        for name in ("bodies", "manifolds", "contact_points", "joints", "sorted", "bp_entries", "new_pairs"):
            f = getattr(L, "phxo_world_" + name)
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.phxo_world_stats.restype = C.POINTER(SolveStats)
        L.phxo_world_stats.argtypes = [C.c_void_p]
        L.phxo_world_sweep_tests.restype = C.c_uint64
        L.phxo_world_sweep_tests.argtypes = [C.c_void_p]
        L.phxo_world_point_overflows.restype = C.c_int
        L.phxo_world_point_overflows.argtypes = [C.c_void_p]
        L.phxo_time_impulse_loop.restype = C.c_double
        L.phxo_time_impulse_loop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def ref_lib():
    """The reference's own header-only code (None when the prebuilt shim is absent)."""
    global _ref
    if _ref is None:
        build()
        path = os.path.join(_HERE, "_ref", "libphyx_ref_leaf.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_sizeof.restype = C.c_int
        R.ref_sizeof.argtypes = [C.c_int]
        R.ref_config_enum.restype = C.c_int
        R.ref_config_enum.argtypes = [C.c_int]
        R.ref_radix_float.restype = C.c_uint32
        R.ref_radix_float.argtypes = [C.c_float]
        R.ref_radix_sort3.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        R.ref_pair_hash.restype = C.c_uint32
        R.ref_pair_hash.argtypes = [C.c_uint32, C.c_uint32]
        R.ref_body_init.argtypes = [C.c_void_p] + [C.c_float] * 6
        R.ref_recompute_aabb.argtypes = [C.c_void_p]
        R.ref_update_geom.argtypes = [C.c_void_p]
        R.ref_rotate.argtypes = [C.c_void_p, C.c_float]
        R.ref_coords_rotate.argtypes = [C.c_void_p, C.c_float]
        R.ref_support_points.restype = C.c_int
        R.ref_support_points.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        R.ref_aabb_intersects.restype = C.c_int
        R.ref_aabb_intersects.argtypes = [C.c_void_p, C.c_void_p]
        R.ref_contact_equals.restype = C.c_int
        R.ref_contact_equals.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        R.ref_contact_point_make.argtypes = [C.c_void_p] + [C.c_float] * 6 + [C.c_void_p, C.c_void_p]
        R.ref_project_point_to_line.argtypes = [C.c_float] * 8 + [C.c_void_p]
        R.ref_pairset_insert_run.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        R.ref_flipsign1.restype = C.c_float
        R.ref_flipsign1.argtypes = [C.c_float, C.c_float]
        R.ref_max1.restype = C.c_float
        R.ref_max1.argtypes = [C.c_float, C.c_float]
        for name in ("ref_simd4_lane0", "ref_simd8_lane0"):
            if hasattr(R, name):
                getattr(R, name).restype = C.c_float
                getattr(R, name).argtypes = [C.c_int, C.c_float, C.c_float]
        _ref = R
    return _ref


ARITH_SOURCE, ARITH_FUSED = 0, 1


def set_arith(fused):
    """The sweeps' arithmetic form of every later oracle solve (phx_oracle.c mul_add / mul_sub): ARITH_SOURCE = rounded product +
    rounded sum, ARITH_FUSED = one fmaf per multiply-add pair.  Process-wide; returns the previous form."""
    prev = lib().phxo_get_arith()
    lib().phxo_set_arith(int(fused))
    return prev


def get_arith():
    return lib().phxo_get_arith()


class OracleWorld:
    """World::Update restated on the CPU (ref: World.cpp:19-37)."""

    def __init__(self, gravity=-200.0):
        self.L = lib()
        self.h = C.c_void_p(self.L.phxo_world_create())
        self.L.phxo_world_set_gravity(self.h, gravity)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.phxo_world_destroy(self.h)
            self.h = None

    def add_body(self, px, py, angle, sx, sy, static=False):
        i = self.L.phxo_world_add_body(self.h, px, py, angle, sx, sy)
        if static:
            b = self.bodies()
            b["inv_mass"][i] = 0.0
            b["inv_inertia"][i] = 0.0
        return i

    def add_scene(self, scene):
        pinned = scene.get("pinned")                 # invMass = 0 only (ref: main.cpp:176-177): set after the loop below
        for k in range(len(scene["px"])):
            self.add_body(float(scene["px"][k]), float(scene["py"][k]), float(scene["angle"][k]),
                          float(scene["sx"][k]), float(scene["sy"][k]), bool(scene["static"][k]))
        if pinned is not None and np.any(pinned):
            self.bodies()["inv_mass"][np.asarray(pinned, dtype=bool)] = 0.0      # the constructor's invInertia stays

    def _view(self, getter, dtype):
        n = C.c_int(0)
        ptr = getter(self.h, C.byref(n))
        if n.value == 0 or not ptr:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n.value * dtype.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n.value)

    # live views into the oracle's own arrays (invalidated by the next update)
    def bodies(self):
        return self._view(self.L.phxo_world_bodies, body_dtype)

    def manifolds(self):
        return self._view(self.L.phxo_world_manifolds, manifold_dtype)

    def contact_points(self):
        return self._view(self.L.phxo_world_contact_points, contact_point_dtype)

    def joints(self):
        return self._view(self.L.phxo_world_joints, joint_dtype)

    def sorted_entries(self):
        return self._view(self.L.phxo_world_sorted, sort_entry_dtype)

    def bp_entries(self):
        return self._view(self.L.phxo_world_bp_entries, bp_entry_dtype)

    def new_pairs(self):
        n = C.c_int(0)
        ptr = self.L.phxo_world_new_pairs(self.h, C.byref(n))
        if n.value == 0 or not ptr:
            return np.zeros((0, 2), dtype=np.uint32)
        buf = (C.c_char * (n.value * 8)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint32, count=2 * n.value).reshape(-1, 2)

    def stats(self):
        return self.L.phxo_world_stats(self.h).contents

    def sweep_tests(self):
        return int(self.L.phxo_world_sweep_tests(self.h))

    def update(self, dt=1.0 / 60.0, solve_mode=SOLVE_SCALAR, island_mode=ISLAND_SINGLE, contact_iters=15, penetration_iters=15):
        self.L.phxo_world_update(self.h, dt, solve_mode, island_mode, contact_iters, penetration_iters)

    def pre_solve(self, dt=1.0 / 60.0):
        self.L.phxo_world_pre_solve(self.h, dt)

    def solve_and_integrate(self, dt=1.0 / 60.0, solve_mode=SOLVE_SCALAR, island_mode=ISLAND_SINGLE, contact_iters=15, penetration_iters=15):
        self.L.phxo_world_solve_and_integrate(self.h, dt, solve_mode, island_mode, contact_iters, penetration_iters)

    def integrate_position(self, dt=1.0 / 60.0):
        self.L.phxo_world_integrate_position(self.h, dt)


def solver_solve(bodies, cps, joints, solve_mode, island_mode, contact_iters, penetration_iters):
    """Solver::SolveJoints in the reference's own order. Mutates bodies/joints. Returns (order, stats)."""
    L = lib()
    st = SolveStats()
    cap = len(joints) + (len(joints) // 256 + 2) * 8 + 16
    order = np.full(cap, -1, dtype=np.int32)
    L.phxo_solver_solve(_p(bodies), len(bodies), _p(cps), _p(joints), len(joints), solve_mode, island_mode,
                        contact_iters, penetration_iters, _p(order), cap, C.byref(st))
    return order, st


def solver_solve_ordered(bodies, cps, joints, order, colour_offsets, contact_iters, penetration_iters, stag_mode=STAG_SEQUENTIAL):
    L = lib()
    st = SolveStats()
    order = np.ascontiguousarray(order, dtype=np.int32)
    co = None if colour_offsets is None else np.ascontiguousarray(colour_offsets, dtype=np.int32)
    L.phxo_solver_solve_ordered(_p(bodies), len(bodies), _p(cps), _p(joints), len(joints), _p(order),
                                _p(co) if co is not None else None, 0 if co is None else len(co) - 1,
                                contact_iters, penetration_iters, stag_mode, C.byref(st))
    return st


def solver_solve_grouped(bodies, cps, joints, order, colour_offsets, group_offsets, contact_iters, penetration_iters,
                         stag_mode=STAG_COLOUR_SYNC, fp16_groups=0):
    """The HIP path's island-aware schedule replayed sequentially: groups are independent islands.
    fp16_groups > 0 models the fp16 body-state ablation for the first that many groups."""
    L = lib()
    st = SolveStats()
    order = np.ascontiguousarray(order, dtype=np.int32)
    co = np.ascontiguousarray(colour_offsets, dtype=np.int32)
    go = np.ascontiguousarray(group_offsets, dtype=np.int32)
    L.phxo_solver_solve_grouped_fp16(_p(bodies), len(bodies), _p(cps), _p(joints), len(joints), _p(order), _p(co), len(co) - 1,
                                     _p(go), len(go) - 1, contact_iters, penetration_iters, stag_mode, fp16_groups, C.byref(st))
    return st


def refresh_joint(bodies, cps, joint_record):
    out = np.zeros(30, dtype=np.float32)
    j = np.array([joint_record], dtype=joint_dtype)
    lib().phxo_refresh_joint(_p(bodies), _p(cps), _p(j), _p(out))
    return out


def broadphase_build(bodies):
    n = len(bodies)
    keys = np.zeros(n, dtype=sort_entry_dtype)
    srt = np.zeros(n, dtype=sort_entry_dtype)
    ent = np.zeros(n, dtype=bp_entry_dtype)
    lib().phxo_broadphase_build(_p(bodies), n, _p(keys), _p(srt), _p(ent))
    return keys, srt, ent


def sweep_candidates(entries, cap=None):
    n = len(entries)
    tests = C.c_uint64(0)
    if cap is None:
        cap = int(lib().phxo_sweep_candidates(_p(entries), n, None, 0, C.byref(tests)))
    pairs = np.zeros((max(cap, 1), 2), dtype=np.uint32)
    cnt = int(lib().phxo_sweep_candidates(_p(entries), n, _p(pairs), cap, C.byref(tests)))
    return pairs[:min(cnt, cap)], cnt, int(tests.value)


def time_impulse_loop(bodies, cps, joints, iters, threads):
    visits = C.c_int64(0)
    b = bodies.copy()
    j = joints.copy()
    sec = lib().phxo_time_impulse_loop(_p(b), len(b), _p(cps), _p(j), len(j), iters, threads, C.byref(visits))
    return sec, int(visits.value)


# ---- the CPU baseline (cpu_baseline.c): 8-wide AVX2-order restatement, built with the reference's flags ("fast") and with
#      the oracle's strict flags ("strict", bit-comparable with phxo_solver_solve in AVX2 mode) -------------------------------
class BaselinePhases(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("prepare_bodies", "prepare_indices", "prepare_joints", "refresh", "prestep", "impulse",
                                          "displacement", "finish", "total")] + \
               [("impulse_iterations", C.c_int32), ("displacement_iterations", C.c_int32), ("group_offset", C.c_int32), ("threads", C.c_int32),
                ("joint_visits", C.c_int64)]


class BaselineBroadphasePhases(C.Structure):
    _fields_ = [("update_broadphase", C.c_double), ("update_pairs", C.c_double), ("candidate_tests", C.c_int64),
                ("overlapping_pairs", C.c_int64), ("threads", C.c_int32), ("reps", C.c_int32)]


_baseline = {}


def baseline_lib(kind="fast"):
    """kind: 'fast' (-O3 -ffast-math -mavx2 -mfma, the reference's Makefile flags) or 'strict' (the oracle's flags)."""
    if kind not in _baseline:
        so = os.path.join(_HERE, "libcpubaseline_%s.so" % kind)
        if os.environ.get("PHX_ORACLE_SAN") == "1":
            so = os.path.join(_HERE, "libcpubaseline_%s_asan.so" % kind)
        src = os.path.join(_HERE, "cpu_baseline.c")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            subprocess.check_call(["make", "-C", _HERE, os.path.basename(so)], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.phxb_solve.restype = C.c_int
        L.phxb_solve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(BaselinePhases)]
        L.phxb_broadphase.restype = C.c_int
        L.phxb_broadphase.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(BaselineBroadphasePhases)]
        _baseline[kind] = L
    return _baseline[kind]


def baseline_solve(bodies, cps, joints, contact_iters, penetration_iters, threads=1, sloppy=False, kind="fast"):
    """Solver::SolveJoints<8> (Single, or Single Sloppy with `threads` workers) IN PLACE on bodies / joints; returns the phases."""
    ph = BaselinePhases()
    baseline_lib(kind).phxb_solve(_p(bodies), len(bodies), _p(cps), _p(joints), len(joints), contact_iters, penetration_iters,
                                  threads, 1 if sloppy else 0, C.byref(ph))
    return ph


def baseline_broadphase(bodies, threads=1, reps=3, kind="fast"):
    ph = BaselineBroadphasePhases()
    baseline_lib(kind).phxb_broadphase(_p(bodies), len(bodies), threads, reps, C.byref(ph))
    return ph
