"""Host-side mirror of the reference's interface for the hot path, over the C ABI.

Names and argument meaning follow the reference (ref = /root/reference/src):
  Configuration  <- Configuration.h:3-23      (solveMode, islandMode, contactIterationsCount, penetrationIterationsCount)
  Solver         <- Solver.h:47-128           (SolveJoints, islandCount, islandMaxSize)
  Collider       <- Collider.h:24-66          (UpdateBroadphase + UpdatePairs, broadphase[], broadphaseSort[1])
  World          <- World.h:9-36              (AddBody, Update, bodies, gravity)
Arrays are numpy structured arrays with the reference's exact POD layouts, so the same buffers can be
handed to the oracle in tests/.  All compute happens inside libphyx_amd.so (HIP); nothing here
falls back to numpy or the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Config, SolveStats, BroadphaseStats, BenchResult, check

SOLVE_SCALAR, SOLVE_SSE2, SOLVE_AVX2 = 0, 1, 2
ISLAND_SINGLE, ISLAND_MULTIPLE, ISLAND_SINGLE_SLOPPY, ISLAND_MULTIPLE_SLOPPY = 0, 1, 2, 3

_vec2 = np.dtype([("x", "<f4"), ("y", "<f4")])
rigid_body_dtype = np.dtype([
    ("index", "<u4"), ("geom_size", _vec2), ("geom_xv", _vec2), ("geom_yv", _vec2), ("geom_pos", _vec2),
    ("aabb_min", _vec2), ("aabb_max", _vec2), ("velocity", _vec2), ("acceleration", _vec2),
    ("displacing_velocity", _vec2), ("angular_velocity", "<f4"), ("angular_acceleration", "<f4"),
    ("displacing_angular_velocity", "<f4"), ("inv_mass", "<f4"), ("inv_inertia", "<f4"),
    ("xv", _vec2), ("yv", _vec2), ("pos", _vec2), ("last_iteration", "<i4"), ("last_displacement_iteration", "<i4"),
])
contact_point_dtype = np.dtype([
    ("delta1", _vec2), ("delta2", _vec2), ("normal", _vec2), ("is_merged", "u1"), ("is_newly_created", "u1"),
    ("pad", "u1", (2,)), ("solver_index", "<i4"),
])
manifold_dtype = np.dtype([("body1", "<i4"), ("body2", "<i4"), ("point_count", "<i4"), ("point_index", "<i4")])
contact_joint_dtype = np.dtype([("contact_point_index", "<i4"), ("body1", "<i4"), ("body2", "<i4"),
                                ("normal_acc", "<f4"), ("friction_acc", "<f4")])
broadphase_entry_dtype = np.dtype([("minx", "<f4"), ("maxx", "<f4"), ("centery", "<f4"), ("extenty", "<f4"), ("index", "<u4")])
sort_entry_dtype = np.dtype([("value", "<u4"), ("index", "<u4")])
assert rigid_body_dtype.itemsize == 128 and contact_point_dtype.itemsize == 32
assert manifold_dtype.itemsize == 16 and contact_joint_dtype.itemsize == 20 and broadphase_entry_dtype.itemsize == 20


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _contig(a, dtype):
    a = np.asarray(a)
    if a.dtype != dtype or not a.flags["C_CONTIGUOUS"]:
        raise TypeError("expected a C-contiguous array of dtype %s" % (dtype,))
    return a


def device_count():
    return check(_lib.load().phx_device_count())


def device_info(device=0):
    L = _lib.load()
    name = C.create_string_buffer(256)
    cu, lds, hbm = C.c_int(), C.c_int(), C.c_int64()
    check(L.phx_device_info(device, name, 256, C.byref(cu), C.byref(lds), C.byref(hbm)))
    return {"name": name.value.decode(), "compute_units": cu.value, "lds_bytes": lds.value, "hbm_bytes": hbm.value}


XCH_PEER_ERROR, XCH_SERIAL_MISMATCH, XCH_TOPOLOGY_MISMATCH, XCH_BAD_SEGMENT = 1, 2, 4, 8
XCH_HEADER_BYTES = 32


def exchange_layout(group_bodies, group_slots, shard_count):
    """Host-only: the deal of the groups to the ranks (longest processing time first by joint count) and the segment layout
    of the island-sharded exchange (include/phyx_amd.h: phx_exchange_layout) -> (group_offset_words, rank_words, segment_words,
    group_owner)."""
    L = _lib.load()
    gb = np.ascontiguousarray(group_bodies, dtype=np.int32)
    gs = np.ascontiguousarray(group_slots, dtype=np.int32)
    off = np.zeros(max(len(gb), 1), dtype=np.int64)
    own = np.zeros(max(len(gb), 1), dtype=np.int32)
    rw = np.zeros(max(shard_count, 1), dtype=np.int64)
    seg = C.c_int64(0)
    check(L.phx_exchange_layout(_ptr(gb), _ptr(gs), len(gb), shard_count, _ptr(own), _ptr(off), _ptr(rw), C.byref(seg)))
    return off[:len(gb)], rw, seg.value, own[:len(gb)]


def schedule_priority(priority_id, joint_index, lower_body=0):
    """Colouring priority of a joint (higher = coloured earlier); see include/phyx_amd.h."""
    return int(_lib.load().phx_schedule_priority(int(priority_id), int(joint_index), int(lower_body)))


def schedule_colours(body1, body2, is_static, priority_ids=None):
    """Host-only: colour classes of body-disjoint joints (this backend's PrepareIndices, ref: Solver.cpp:217-273).
    priority_ids: per-joint priority id (the solver uses contactPointIndex); the joint index if None."""
    L = _lib.load()
    pid = None if priority_ids is None else np.ascontiguousarray(priority_ids, dtype=np.int32)
    b1 = np.ascontiguousarray(body1, dtype=np.int32)
    b2 = np.ascontiguousarray(body2, dtype=np.int32)
    st = np.ascontiguousarray(is_static, dtype=np.uint8)
    order = np.zeros(max(len(b1), 1), dtype=np.int32)
    offs = np.zeros(len(b1) + 2, dtype=np.int32)
    nc = C.c_int32(0)
    check(L.phx_schedule_colours(_ptr(b1), _ptr(b2), len(b1), _ptr(st), len(st), None if pid is None else _ptr(pid), _ptr(order), _ptr(offs), len(offs), C.byref(nc)))
    return order[:len(b1)], offs[:nc.value + 1]


def schedule_groups(body1, body2, is_static, priority_ids=None, lanes=256, body_cap=768):
    """Host-only: the island-mode schedule as workgroup-sized LDS groups (include/phyx_amd.h phx_schedule_groups) -> dict with
    order, colour_offsets, group_offsets, group_first_colour, lds_groups, unit_lane, unit_leader_slot."""
    L = _lib.load()
    pid = None if priority_ids is None else np.ascontiguousarray(priority_ids, dtype=np.int32)
    b1 = np.ascontiguousarray(body1, dtype=np.int32)
    b2 = np.ascontiguousarray(body2, dtype=np.int32)
    st = np.ascontiguousarray(is_static, dtype=np.uint8)
    n = len(b1)
    order = np.zeros(max(n, 1), dtype=np.int32)
    offs = np.zeros(n + 2, dtype=np.int32)
    goff = np.zeros(n + 3, dtype=np.int32); gfc = np.zeros(n + 3, dtype=np.int32)
    ulane = np.zeros(max(n, 1), dtype=np.int32); uslot = np.zeros(max(n, 1), dtype=np.int32)
    nc = C.c_int32(0); lg = C.c_int32(0)
    nu = check(L.phx_schedule_groups(_ptr(b1), _ptr(b2), n, _ptr(st), len(st), None if pid is None else _ptr(pid), int(lanes), int(body_cap), _ptr(order), _ptr(offs),
                                     len(offs), C.byref(nc), _ptr(goff), _ptr(gfc), len(goff), C.byref(lg), _ptr(ulane), _ptr(uslot)))
    ng = lg.value + (1 if lg.value == 0 or goff[lg.value] < n else 0)
    return dict(order=order[:n], colour_offsets=offs[:nc.value + 1], lds_groups=lg.value, group_offsets=goff[:ng + 1], group_first_colour=gfc[:ng + 1],
                unit_lane=ulane[:nu], unit_leader_slot=uslot[:nu])


def schedule_islands(body1, body2, is_static):
    """Host-only: GatherIslands semantics (ref: Solver.cpp:285-454) -> (joint_island, island_size)."""
    L = _lib.load()
    b1 = np.ascontiguousarray(body1, dtype=np.int32)
    b2 = np.ascontiguousarray(body2, dtype=np.int32)
    st = np.ascontiguousarray(is_static, dtype=np.uint8)
    ji = np.zeros(max(len(b1), 1), dtype=np.int32)
    sz = np.zeros(len(b1) + 1, dtype=np.int32)
    n = check(L.phx_schedule_islands(_ptr(b1), _ptr(b2), len(b1), _ptr(st), len(st), _ptr(ji), _ptr(sz), len(sz)))
    return ji[:len(b1)], sz[:n]


class Configuration:
    """ref: Configuration.h:20-23; defaults are the demo's (main.cpp:262-263,348) with the scalar mode."""

    def __init__(self, solveMode=SOLVE_SCALAR, islandMode=ISLAND_SINGLE, contactIterationsCount=15, penetrationIterationsCount=15):
        self.solveMode = solveMode
        self.islandMode = islandMode
        self.contactIterationsCount = contactIterationsCount
        self.penetrationIterationsCount = penetrationIterationsCount

    def _c(self):
        return Config(self.solveMode, self.islandMode, self.contactIterationsCount, self.penetrationIterationsCount)


class DeviceBuffer:
    """A raw HBM allocation of `nbytes` (exchange buffers of a sharded solve when the caller has no torch tensors)."""

    def __init__(self, nbytes, device=0):
        self.L = _lib.load()
        self.device = device
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(self.L.phx_device_malloc(device, max(self.nbytes, 1), C.byref(p)))
        self.ptr = p

    def address(self, offset=0):
        return C.c_void_p(self.ptr.value + int(offset))

    def to_host(self, nbytes=None, offset=0):
        n = self.nbytes - offset if nbytes is None else int(nbytes)
        out = np.zeros(n, dtype=np.uint8)
        if n:
            check(self.L.phx_memcpy_d2h(self.device, _ptr(out), self.address(offset), n))
        return out

    def from_host(self, data, offset=0):
        a = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        if len(a):
            check(self.L.phx_memcpy_h2d(self.device, self.address(offset), _ptr(a), len(a)))

    def copy_from(self, other, nbytes, dst_offset=0, src_offset=0, stream=None):
        """Device-to-device copy; with `stream` (a hipStream_t address) it is queued on that stream instead of blocking."""
        if not nbytes:
            return
        if stream is None:
            check(self.L.phx_memcpy_d2d(self.device, self.address(dst_offset), other.address(src_offset), int(nbytes)))
        else:
            check(self.L.phx_memcpy_d2d_on(self.device, self.address(dst_offset), other.address(src_offset), int(nbytes), C.c_void_p(int(stream))))

    def free(self):
        if self.ptr:
            self.L.phx_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceArray:
    """A raw HBM allocation holding a copy of a host array (inputs resident before a timed region)."""

    def __init__(self, host_array, device=0):
        self.L = _lib.load()
        self.device = device
        self.dtype = host_array.dtype
        self.count = len(host_array)
        self.nbytes = int(host_array.nbytes)
        p = C.c_void_p()
        check(self.L.phx_device_malloc(device, max(self.nbytes, 1), C.byref(p)))
        self.ptr = p
        if self.nbytes:
            check(self.L.phx_memcpy_h2d(device, self.ptr, _ptr(np.ascontiguousarray(host_array)), self.nbytes))

    def to_host(self):
        out = np.zeros(self.count, dtype=self.dtype)
        if self.nbytes:
            check(self.L.phx_memcpy_d2h(self.device, _ptr(out), self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.L.phx_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Solver:
    """Device replacement of Solver (ref: Solver.h:47-128)."""

    def __init__(self, device=0, _handle=None):
        self.L = _lib.load()
        self.device = device
        self._owned = _handle is None
        if _handle is None:
            h = C.c_void_p()
            check(self.L.phx_solver_create(C.byref(h), device))
            self.h = h
        else:
            self.h = C.c_void_p(_handle)
        self.islandCount = 0
        self.islandMaxSize = 0

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "h", None):
            self.L.phx_solver_destroy(self.h)
            self.h = None

    def SolveJoints(self, bodies, contactPoints, contactJoints, configuration):
        """ref: Solver.h:54 — solves in place on the host arrays (velocities + accumulated impulses)."""
        b = _contig(bodies, rigid_body_dtype)
        cp = _contig(contactPoints, contact_point_dtype)
        j = _contig(contactJoints, contact_joint_dtype)
        cfg = configuration._c()
        check(self.L.phx_solver_solve(self.h, _ptr(b), len(b), _ptr(cp), len(cp), _ptr(j), len(j), C.byref(cfg)))
        st = self.stats()
        self.islandCount, self.islandMaxSize = st.island_count, st.island_max_size
        return st

    def SolveJointsDevice(self, d_bodies, d_contact_points, d_joints, configuration):
        cfg = configuration._c()
        check(self.L.phx_solver_solve_device(self.h, d_bodies.ptr, d_bodies.count, d_contact_points.ptr, d_contact_points.count,
                                             d_joints.ptr, d_joints.count, C.byref(cfg)))

    def synchronize(self):
        check(self.L.phx_solver_synchronize(self.h))

    def set_body_state_bits(self, bits):
        """32 = fp32 solver-side body state (default, the reference's); 16 = the fp16 ablation of BASELINE config 5."""
        check(self.L.phx_solver_set_body_state_bits(self.h, bits))

    def set_schedule_reuse(self, on):
        """False: rebuild the schedule on every solve (the reference rebuilds its grouping every call)."""
        check(self.L.phx_solver_set_schedule_reuse(self.h, 1 if on else 0))

    def set_trace(self, on=True, waves=True):
        """Phase stamps of the island kernel's workgroups (island_trace); waves=True also counts every wave's class-step cycles
        (wave_trace), which slows the sweeps by ~15 %."""
        check(self.L.phx_solver_set_trace(self.h, (2 if waves else 1) if on else 0))

    def island_trace(self):
        """(groups, 8) uint64: phase stamps of the island kernel's workgroups in the last solve (set_trace first)."""
        n = C.c_int32(0)
        check(self.L.phx_solver_get_island_trace(self.h, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 8), dtype=np.uint64)
        check(self.L.phx_solver_get_island_trace(self.h, _ptr(out), n.value, C.byref(n)))
        return out[:n.value]

    def wave_trace(self):
        """(groups, waves, 8) uint64: per-wave colour-step cycles of the last traced solve."""
        n, w = C.c_int32(0), C.c_int32(0)
        check(self.L.phx_solver_get_island_trace(self.h, None, 0, C.byref(n)))
        check(self.L.phx_solver_get_wave_trace(self.h, None, 0, C.byref(w)))
        out = np.zeros((max(n.value, 1), w.value, 8), dtype=np.uint64)
        check(self.L.phx_solver_get_wave_trace(self.h, _ptr(out), out.size, C.byref(w)))
        return out[:n.value]

    def set_shard(self, shard, shard_count):
        """Sweep only the schedule groups g with g % shard_count == shard (multi-GPU island sharding)."""
        check(self.L.phx_solver_set_shard(self.h, shard, shard_count))

    # ---- island-sharded solves: the post-solve exchange (include/phyx_amd.h: phx_solver_exchange_pack) ----
    def set_comm(self, comm):
        """bench(): the all-gather between pack and unpack runs natively on the solver's stream (no step hook needed)."""
        self._comm = comm
        check(self.L.phx_solver_set_comm(self.h, comm.h if comm is not None else None))

    def set_exchange_buffers(self, send_ptr, recv_ptr, segment_capacity_bytes):
        """Caller-owned device buffers (raw addresses): one segment to send, shard_count segments to receive."""
        check(self.L.phx_solver_set_exchange_buffers(self.h, C.c_void_p(int(send_ptr)), C.c_void_p(int(recv_ptr)), int(segment_capacity_bytes)))

    def exchange_pack(self, d_bodies, d_joints, status_word=0):
        seg = C.c_size_t(0)
        check(self.L.phx_solver_exchange_pack(self.h, d_bodies.ptr, d_joints.ptr, int(status_word), C.byref(seg)))
        return seg.value

    def exchange_unpack(self, d_bodies, d_joints):
        check(self.L.phx_solver_exchange_unpack(self.h, d_bodies.ptr, d_joints.ptr))

    def exchange_status(self):
        v = C.c_int32(0)
        check(self.L.phx_solver_exchange_status(self.h, C.byref(v)))
        return v.value

    def exchange_segment_bytes(self):
        return int(self.L.phx_solver_exchange_segment_bytes(self.h))

    def stats(self):
        st = SolveStats()
        check(self.L.phx_solver_get_stats(self.h, C.byref(st)))
        return st

    def schedule(self):
        """(order, colour_offsets) of the last solve: order[k] = joint in slot k."""
        nc = C.c_int32(0)
        check(self.L.phx_solver_get_schedule(self.h, None, 0, None, 0, C.byref(nc)))
        offs = np.zeros(nc.value + 1, dtype=np.int32)
        check(self.L.phx_solver_get_schedule(self.h, None, 0, _ptr(offs), len(offs), C.byref(nc)))
        order = np.zeros(max(int(offs[-1]), 1), dtype=np.int32)
        check(self.L.phx_solver_get_schedule(self.h, _ptr(order), len(order), _ptr(offs), len(offs), C.byref(nc)))
        return order[:int(offs[-1])], offs

    def groups(self):
        """(group_offsets, lds_group_count) of the last solve's schedule."""
        n, lds = C.c_int32(0), C.c_int32(0)
        check(self.L.phx_solver_get_groups(self.h, None, 0, C.byref(n), C.byref(lds)))
        offs = np.zeros(n.value + 1, dtype=np.int32)
        check(self.L.phx_solver_get_groups(self.h, _ptr(offs), len(offs), C.byref(n), C.byref(lds)))
        return offs, lds.value

    def lanes(self):
        """(leader_slot, lane) per unit of the last solve's LDS groups (phx_solver_get_lanes)."""
        n = C.c_int32(0)
        check(self.L.phx_solver_get_lanes(self.h, None, None, 0, C.byref(n)))
        slot = np.zeros(max(n.value, 1), dtype=np.int32); lane = np.zeros(max(n.value, 1), dtype=np.int32)
        check(self.L.phx_solver_get_lanes(self.h, _ptr(slot), _ptr(lane), len(slot), C.byref(n)))
        return slot[:n.value], lane[:n.value]

    def partition(self):
        """(interior_classes, parts, sweep_launches) of the last solve: the partitioned-component path (phx_solver_get_partition)."""
        ki, parts, launches = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        check(self.L.phx_solver_get_partition(self.h, C.byref(ki), C.byref(parts), C.byref(launches)))
        return ki.value, parts.value, launches.value

    def refreshed(self, joint_index):
        out = np.zeros(30, dtype=np.float32)
        check(self.L.phx_solver_get_refreshed(self.h, joint_index, _ptr(out)))
        return out

    def bench_stage(self, d_bodies, d_joints, steps):
        """Make `steps` private copies of the input in HBM NOW (before the caller starts its clock): the next bench() call on the same
        arrays solves copy k in step k instead of restoring a working copy in front of every step."""
        check(self.L.phx_solver_bench_stage(self.h, d_bodies.ptr, d_bodies.count, d_joints.ptr, d_joints.count, steps))

    def bench_checksum(self):
        """64-bit checksum of the results of the last bench() step (velocities, displacing velocities, impulses)."""
        out = C.c_uint64(0)
        check(self.L.phx_solver_bench_checksum(self.h, C.byref(out)))
        return int(out.value)

    def bench(self, d_bodies, d_contact_points, d_joints, configuration, warmup, steps, hook=None):
        """`steps` solves of the same resident input, queued back to back.  hook(step, phase) (optional) runs on the host:
        phase 0 after step `step` has been queued (start the per-step exchange on stream_ptr()), phase 1 when the local
        preparation of step `step` is queued and its sweeps are not (make the stream wait for the previous exchange)."""
        cfg = configuration._c()
        res = BenchResult()
        if hook is None:
            check(self.L.phx_solver_bench(self.h, d_bodies.ptr, d_bodies.count, d_contact_points.ptr, d_contact_points.count,
                                          d_joints.ptr, d_joints.count, C.byref(cfg), warmup, steps, C.byref(res)))
            return res
        failure = []

        def _trampoline(_user, step, phase):
            try:
                hook(int(step), int(phase))
                return 0
            except BaseException as e:          # an exception must not unwind through the C frames
                failure.append(e)
                return 1
        cb = _lib.STEP_HOOK(_trampoline)
        st = self.L.phx_solver_bench_hooked(self.h, d_bodies.ptr, d_bodies.count, d_contact_points.ptr, d_contact_points.count,
                                            d_joints.ptr, d_joints.count, C.byref(cfg), warmup, steps, cb, None, C.byref(res))
        if failure:
            raise failure[0]
        check(st)
        return res

    def stream_ptr(self):
        """The hipStream_t (as an int) all of this handle's work is queued on."""
        return int(self.L.phx_solver_stream(self.h) or 0)


class Comm:
    """Native RCCL communicator (csrc/comm.hip): one per process / GPU.  `unique_id()` on rank 0, hand the 128 bytes to every
    rank out of band, then every rank constructs Comm(id, rank, nranks, device) — a collective call."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        L = _lib.load()
        buf = C.create_string_buffer(Comm.ID_BYTES)
        check(L.phx_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def rccl_version():
        """ncclGetVersion of the RCCL the library resolved (0: none)"""
        return int(_lib.load().phx_comm_rccl_version())

    def __init__(self, unique_id, rank, nranks, device=0):
        self.L = _lib.load()
        assert len(unique_id) == Comm.ID_BYTES
        h = C.c_void_p()
        check(self.L.phx_comm_create(C.byref(h), C.c_char_p(bytes(unique_id)), rank, nranks, device))
        self.h = h
        self.rank, self.size, self.device = rank, nranks, device

    def __del__(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            self.L.phx_comm_destroy(h)
            self.h = C.c_void_p()

    def all_gather(self, send_ptr, recv_ptr, bytes_per_rank, stream_ptr):
        check(self.L.phx_comm_all_gather(self.h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), bytes_per_rank, C.c_void_p(stream_ptr)))

    def barrier(self, stream_ptr=0):
        check(self.L.phx_comm_barrier(self.h, C.c_void_p(stream_ptr)))

    def barrier_async(self, stream_ptr):
        check(self.L.phx_comm_barrier_async(self.h, C.c_void_p(stream_ptr)))

    def async_error(self):
        e = C.c_int32(0)
        check(self.L.phx_comm_async_error(self.h, C.byref(e)))
        return e.value


class Collider:
    """Device replacement of the broadphase half of Collider (ref: Collider.h:28-29, 58-66)."""

    def __init__(self, device=0, _handle=None):
        self.L = _lib.load()
        self._owned = _handle is None
        if _handle is None:
            h = C.c_void_p()
            check(self.L.phx_broadphase_create(C.byref(h), device))
            self.h = h
        else:
            self.h = C.c_void_p(_handle)

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "h", None):
            self.L.phx_broadphase_destroy(self.h)
            self.h = None

    def clear(self):
        check(self.L.phx_broadphase_clear(self.h))

    def UpdateBroadphaseAndPairs(self, bodies):
        """UpdateBroadphase + UpdatePairs; returns the pairs that were new this step, (n,2) uint32 in the
        reference's serial emission order; they are now part of the persistent set."""
        b = _contig(bodies, rigid_body_dtype)
        n = C.c_int32(0)
        check(self.L.phx_broadphase_update(self.h, _ptr(b), len(b), None, 0, C.byref(n)))
        pairs = np.zeros((max(n.value, 1), 2), dtype=np.uint32)
        check(self.L.phx_broadphase_get_new_pairs(self.h, _ptr(pairs), n.value, C.byref(n)))
        return pairs[:n.value]

    def sorted(self, n):
        srt = np.zeros(n, dtype=sort_entry_dtype)
        ent = np.zeros(n, dtype=broadphase_entry_dtype)
        check(self.L.phx_broadphase_get_sorted(self.h, _ptr(srt), _ptr(ent), n))
        return srt, ent

    def erase(self, pairs):
        p = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        check(self.L.phx_broadphase_erase_pairs(self.h, _ptr(p), len(p)))

    def stats(self):
        st = BroadphaseStats()
        check(self.L.phx_broadphase_get_stats(self.h, C.byref(st)))
        return st


class World:
    """Device-backed World (ref: World.h:9-36)."""

    PHASES = ("IntegrateVelocity", "UpdateBroadphase", "UpdatePairs", "UpdateManifolds", "PackManifolds",
              "RefreshContactJoints", "SolveJoints", "IntegratePosition")

    def __init__(self, device=0, gravity=0.0):
        self.L = _lib.load()
        h = C.c_void_p()
        check(self.L.phx_world_create(C.byref(h), device))
        self.h = h
        self.device = device
        self.gravity = gravity
        self.solver = Solver(device, _handle=self.L.phx_world_solver(self.h))
        self.collider = Collider(device, _handle=self.L.phx_world_broadphase(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.phx_world_destroy(self.h)
            self.h = None

    @property
    def gravity(self):
        return self._gravity

    @gravity.setter
    def gravity(self, g):
        self._gravity = float(g)
        check(self.L.phx_world_set_gravity(self.h, self._gravity))

    def AddBody(self, pos, angle, size, static=False):
        """ref: World.cpp:11-17 — pos=(x,y), size = half extents; returns the body index."""
        i = check(self.L.phx_world_add_body(self.h, pos[0], pos[1], angle, size[0], size[1]))
        if static:
            check(self.L.phx_world_set_body_static(self.h, i))
        return i

    def set_inverse_mass(self, body, inv_mass, inv_inertia):
        """body->invMass / invInertia (the demo pins shelves with invMass = 0 only, ref: main.cpp:176-177)."""
        check(self.L.phx_world_set_body_inverse_mass(self.h, int(body), float(inv_mass), float(inv_inertia)))

    def add_scene(self, scene):
        pinned = scene.get("pinned")                 # invMass = 0 only: infinitely heavy but free to rotate
        for k in range(len(scene["px"])):
            i = self.AddBody((float(scene["px"][k]), float(scene["py"][k])), float(scene["angle"][k]),
                             (float(scene["sx"][k]), float(scene["sy"][k])), bool(scene["static"][k]))
            if pinned is not None and pinned[k]:
                sx, sy = float(scene["sx"][k]), float(scene["sy"][k])
                mass = np.float32(1e-5) * (np.float32(sx) * np.float32(sy))
                inertia = mass * (np.float32(sx) * np.float32(sx) + np.float32(sy) * np.float32(sy))
                self.set_inverse_mass(i, 0.0, float(np.float32(1.0) / inertia))

    def set_shard(self, shard, shard_count):
        check(self.L.phx_world_set_shard(self.h, shard, shard_count))

    def Update(self, dt, configuration):
        cfg = configuration._c()
        check(self.L.phx_world_update(self.h, dt, C.byref(cfg)))

    def StepBegin(self, dt, configuration):
        """First half of a SHARDED world's step: everything up to and including SolveJoints of this rank's groups, then
        the pack.  Returns the segment size in bytes every rank must all-gather (send buffer -> recv buffer)."""
        cfg = configuration._c()
        seg = C.c_size_t(0)
        check(self.L.phx_world_step_begin(self.h, dt, C.byref(cfg), C.byref(seg)))
        return seg.value

    def StepEnd(self, dt):
        """Second half: scatter the other ranks' results, IntegratePosition."""
        check(self.L.phx_world_step_end(self.h, dt))

    def set_comm(self, comm):
        """Attach a native communicator (Comm): this world becomes rank comm.rank of comm.size and owns its exchange buffers."""
        self._comm = comm
        check(self.L.phx_world_set_comm(self.h, comm.h if comm is not None else None))

    def StepSharded(self, dt, configuration):
        """World::Update of a sharded world in one library call: step_begin, ncclAllGather on the world's stream, step_end."""
        cfg = configuration._c()
        check(self.L.phx_world_step_sharded(self.h, dt, C.byref(cfg)))

    def check_exchange(self):
        check(self.L.phx_world_check_exchange(self.h))

    def stream_ptr(self):
        return int(self.L.phx_world_stream(self.h) or 0)

    def PreSolve(self, dt):
        """Everything of World::Update that precedes Solver::SolveJoints (ref: World.cpp:25-32)."""
        check(self.L.phx_world_pre_solve(self.h, dt))

    def FinishStep(self, dt, configuration):
        """Solver::SolveJoints + IntegratePosition (ref: World.cpp:34-36)."""
        cfg = configuration._c()
        check(self.L.phx_world_finish_step(self.h, dt, C.byref(cfg)))

    def counts(self):
        v = [C.c_int32() for _ in range(4)]
        check(self.L.phx_world_counts(self.h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def _get(self, fn, dtype, n):
        out = np.zeros(n, dtype=dtype)
        check(fn(self.h, _ptr(out), n))
        return out

    @property
    def bodies(self):
        return self._get(self.L.phx_world_get_bodies, rigid_body_dtype, self.counts()[0])

    @property
    def manifolds(self):
        return self._get(self.L.phx_world_get_manifolds, manifold_dtype, self.counts()[1])

    @property
    def contactPoints(self):
        return self._get(self.L.phx_world_get_contact_points, contact_point_dtype, self.counts()[2])

    @property
    def contactJoints(self):
        return self._get(self.L.phx_world_get_joints, contact_joint_dtype, self.counts()[3])

    def state(self):
        """(bodies, manifolds, contact points, joints): everything a world carries from one step to the next."""
        return self.bodies, self.manifolds, self.contactPoints, self.contactJoints

    def set_state(self, bodies, manifolds, contact_points, joints):
        """Restore a saved state() (phx_world_set_state): the world then steps exactly like the one the state was taken from."""
        b = np.ascontiguousarray(bodies, dtype=rigid_body_dtype); m = np.ascontiguousarray(manifolds, dtype=manifold_dtype)
        c = np.ascontiguousarray(contact_points, dtype=contact_point_dtype); j = np.ascontiguousarray(joints, dtype=contact_joint_dtype)
        check(self.L.phx_world_set_state(self.h, _ptr(b), len(b), _ptr(m), len(m), _ptr(c), len(c), _ptr(j), len(j)))

    def sync(self):
        """Wait for the queued step (Update returns once the step is queued; getters synchronise on their own)."""
        check(self.L.phx_world_synchronize(self.h))

    def x_extent(self):
        """(min x, max x) over the dynamic bodies' AABBs, reduced on the device (phx_world_x_extent)."""
        out = np.zeros(2, dtype=np.float32)
        check(self.L.phx_world_x_extent(self.h, _ptr(out)))
        return float(out[0]), float(out[1])

    def reslab(self, global_index, scene_size, bounds, margin=1.0, rank=0, size=1, comm=None, all_gather=None, all_reduce_max=None):
        """phx_world_reslab: the collective hand-over of an ownership-sharded world's bodies (every rank calls it at the same step).
        global_index: scene index of every body of this world.  comm: a phyx_amd.Comm (collectives on device buffers through RCCL), or
        the two host callables all_gather(send: uint8 array) -> uint8 array of size * len(send) and all_reduce_max(int) -> int.
        Returns (moved, global_index, (lo, hi)): moved = somebody changed owner and this world was rebuilt for its new slab."""
        from ._lib import SlabTransport, SLAB_ALL_GATHER, SLAB_ALL_REDUCE_MAX
        gi = np.zeros(int(scene_size), dtype=np.int64)
        gi[:len(global_index)] = np.asarray(global_index, dtype=np.int64)
        count = C.c_int32(len(global_index))
        b = np.asarray([bounds[0], bounds[1]], dtype=np.float64)
        moved = C.c_int32(0)
        errors = []

        def gather_cb(user, send, recv, nbytes):
            try:
                mine = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,)).copy()
                out = np.ascontiguousarray(all_gather(mine), dtype=np.uint8).reshape(-1)
                if out.size != nbytes * size:
                    raise RuntimeError("all_gather returned %d bytes, expected %d" % (out.size, nbytes * size))
                C.memmove(recv, out.ctypes.data, out.size)
                return 0
            except Exception as e:          # (an exception must not unwind through the C frames)
                errors.append(e)
                return 1

        def max_cb(user, value):
            try:
                value[0] = int(all_reduce_max(int(value[0])))
                return 0
            except Exception as e:
                errors.append(e)
                return 1
        tp = SlabTransport()
        tp.rank, tp.size = int(rank), int(size)
        tp.comm = comm.h if comm is not None else None
        g_cb = SLAB_ALL_GATHER(gather_cb) if all_gather is not None else SLAB_ALL_GATHER()
        m_cb = SLAB_ALL_REDUCE_MAX(max_cb) if all_reduce_max is not None else SLAB_ALL_REDUCE_MAX()
        tp.all_gather, tp.all_reduce_max, tp.user = g_cb, m_cb, None
        st = self.L.phx_world_reslab(self.h, C.byref(tp), _ptr(gi), len(gi), C.byref(count), int(scene_size), float(margin), _ptr(b), C.byref(moved))
        if errors:
            raise errors[0]
        check(st)
        return bool(moved.value), gi[:count.value].copy(), (float(b[0]), float(b[1]))

    def build_counts(self):
        """(rebuilds whose components and bins came from the manifolds, rebuilds from the joints) so far (phx_world_build_counts)."""
        out = np.zeros(2, dtype=np.int64)
        check(self.L.phx_world_build_counts(self.h, _ptr(out)))
        return int(out[0]), int(out[1])

    def debug_counters(self):
        """{deferred_packs, deferred_pack_retries, solve_replays, dropped_points} (phx_world_debug_counters)."""
        out = np.zeros(4, dtype=np.int64)
        check(self.L.phx_world_debug_counters(self.h, _ptr(out)))
        return dict(zip(("deferred_packs", "deferred_pack_retries", "solve_replays", "dropped_points"), (int(x) for x in out)))

    def set_phase_timing(self, on=True):
        """Per-phase host timers (phase_ms) cost one stream synchronisation per phase; off by default."""
        check(self.L.phx_world_set_phase_timing(self.h, 1 if on else 0))

    def phase_ms(self):
        out = np.zeros(8, dtype=np.float64)
        check(self.L.phx_world_get_phase_ms(self.h, _ptr(out)))
        return dict(zip(self.PHASES, out.tolist()))
