"""One process per GPU: rendezvous, the per-step exchange and max-over-ranks reduction for bench.py.

The hot path shards by island: every rank steps a replica of the world and solves its own groups; ONE all-gather per
step carries each rank's results (and its status word) to every other rank — it is also the per-step barrier
(BASELINE.json north_star).  The layout, the pack / unpack kernels AND the transport live in the C library
(csrc/exchange.h, csrc/comm.hip): with backend "rccl" (the default on a GPU node) the all-gather is ncclAllGather called
by the library on the solver's stream — no Python in a step — and torch.distributed (gloo, CPU) only does the rendezvous
(hands rank 0's communicator id to every rank) and the max-over-ranks reductions of the timing.  Two more transports are
kept: "nccl" = torch.distributed's ProcessGroupNCCL driven from a per-step Python hook (round 2's path), and "gloo" =
segments staged through the host (CPU tests, several ranks sharing one GPU).
"""
import os

import numpy as np


class ExchangeError(RuntimeError):
    """A peer reported a failure, is at another step, or solved another topology (PHX_XCH_* bits in .status)."""

    def __init__(self, status):
        names = [n for b, n in ((1, "peer error"), (2, "step serial mismatch"), (4, "topology mismatch: replicas diverged"),
                                (8, "segment never written")) if status & b]
        super().__init__("island-sharded exchange failed: %s (status %d)" % (", ".join(names) or "?", status))
        self.status = status


class Exchange:
    """Buffers + transport of the island-sharded exchange for one solver handle.

    all_gather(segment_bytes) moves the first `segment_bytes` of every rank's send buffer into every rank's recv
    buffer (rank r at byte offset r * segment_bytes), ordered on the solver's stream `stream_ptr`:
      nccl   torch uint8 tensors, dist.all_gather_into_tensor under an ExternalStream of that stream: RCCL runs behind
             the pack kernels and in front of the unpack kernels, the host never waits;
      gloo   staged through the host (d2h on the stream, gloo all_gather, h2d on the stream) — functional path for CPU
             rendezvous and for ranks that share one GPU;
      single one rank: a device-to-device copy on the stream.
    """

    def __init__(self, group, solver, capacity_bytes, device=0):
        from . import api
        self.group, self.solver, self.device = group, solver, device
        self.n = group.world_size
        self.capacity = (int(capacity_bytes) + 255) // 256 * 256
        self.stream_ptr = solver.stream_ptr()
        self.backend = getattr(group, "backend", "single")
        self.L = api._lib.load()
        self.comm = getattr(group, "comm", None) if self.backend == "rccl" else None
        if self.backend == "nccl":
            torch = group.torch
            self.send = torch.zeros(self.capacity, dtype=torch.uint8, device=group.device)
            self.recv = torch.zeros(self.capacity * self.n, dtype=torch.uint8, device=group.device)
            torch.cuda.synchronize(group.device)
            self.ext = torch.cuda.ExternalStream(self.stream_ptr, device=group.device)
            send_ptr, recv_ptr = self.send.data_ptr(), self.recv.data_ptr()
        else:
            self.send = api.DeviceBuffer(self.capacity, device)
            self.recv = api.DeviceBuffer(self.capacity * self.n, device)
            send_ptr, recv_ptr = self.send.ptr.value, self.recv.ptr.value
        self.send_ptr, self.recv_ptr = send_ptr, recv_ptr
        solver.set_exchange_buffers(send_ptr, recv_ptr, self.capacity)
        if self.comm is not None:
            solver.set_comm(self.comm)      # Solver.bench: pack -> ncclAllGather -> unpack inside the library, no step hook

    @staticmethod
    def capacity_for(body_count, joint_count):
        """Upper bound of a segment: one rank owning every group (6 floats per body, 2 per joint, header, padding)."""
        # (a body may appear in one dynamic group; static bodies appear in every group that touches them: <= 2 bodies per joint)
        n = 32 + 24 * (int(body_count) + 2 * int(joint_count)) + 8 * int(joint_count) + 16 * (int(joint_count) + 2) + 256
        return (n + 65535) // 65536 * 65536          # (segments are padded to 64 KB, csrc/exchange.h)

    def all_gather(self, segment_bytes):
        seg = int(segment_bytes)
        if seg > self.capacity:
            raise ValueError("segment of %d bytes exceeds the exchange capacity %d" % (seg, self.capacity))
        if self.comm is not None:
            self.comm.all_gather(self.send_ptr, self.recv_ptr, seg, self.stream_ptr)
            return
        if self.backend == "nccl":
            torch, dist = self.group.torch, self.group.dist
            with torch.cuda.stream(self.ext):
                dist.all_gather_into_tensor(self.recv[: seg * self.n], self.send[:seg])
            return
        import ctypes as C
        stream = C.c_void_p(self.stream_ptr)
        if self.backend == "single" or self.n == 1:
            from .api import check
            check(self.L.phx_memcpy_d2d_on(self.device, self.recv.address(0), self.send.address(0), seg, stream))
            return
        # gloo: host staging, ordered on the solver's stream by the blocking stream copies
        from .api import check
        mine = np.zeros(seg, dtype=np.uint8)
        check(self.L.phx_memcpy_d2h_on(self.device, mine.ctypes.data_as(C.c_void_p), self.send.address(0), seg, stream))
        allb = self.group.all_gather_bytes(mine)
        check(self.L.phx_memcpy_h2d_on(self.device, self.recv.address(0), allb.ctypes.data_as(C.c_void_p), seg * self.n, stream))

    def hook(self):
        """Step hook for Solver.bench: phase 2 = results packed on the stream, run the all-gather now.
        (None with the native transport: the library runs the collective itself.)"""
        if self.comm is not None:
            return None

        def hook(step, phase):
            if phase == 2:
                self.all_gather(self.solver.exchange_segment_bytes())
        return hook

    def check(self):
        """Synchronises; raises ExchangeError on every rank that saw an inconsistent exchange."""
        st = self.solver.exchange_status()
        if st:
            raise ExchangeError(st)


def step_sharded(world, dt, configuration, exchange):
    """One World::Update of an island-sharded world (ref: World.cpp:19-37 on every rank's replica): solve this rank's
    groups, all-gather everyone's results, scatter them, integrate.  Before the collective the ranks AGREE (one small
    all-reduce: Group.agree) on whether anyone failed in this step and on the segment size: a failed or diverged step ends
    there on every rank with an error, and nobody enters an all-gather whose byte count its peers do not share."""
    from ._lib import PhxError
    err, seg = None, 0
    try:
        seg = world.StepBegin(dt, configuration)
    except PhxError as e:
        err = e
    worst, lo, hi = exchange.group.agree(1 if err is not None else 0, seg)
    if err is not None:
        raise err
    if worst:
        raise ExchangeError(1)
    if lo != hi:
        raise ExchangeError(4)
    exchange.all_gather(seg)
    world.StepEnd(dt)
    exchange.steps = getattr(exchange, "steps", 0) + 1
    if exchange.steps % 16 == 0:                   # a diverged or failed peer must not go unnoticed for long
        exchange.check()


class Single:
    """world_size == 1: no torch import, no collective."""
    rank, world_size, local_rank = 0, 1, 0

    def barrier(self):
        pass

    def step_barrier(self):
        pass

    def step_barrier_value(self, value):
        return int(value)

    def agree(self, status, nbytes):
        return int(status), int(nbytes), int(nbytes)

    def all_gather_bytes(self, mine):
        return np.ascontiguousarray(mine, dtype=np.uint8)

    def reduce_max(self, x):
        return float(x)

    def reduce_sum(self, x):
        return float(x)

    def stream_hook(self, stream_ptr):
        return None

    def exchange(self, solver, capacity_bytes, device=0):
        return Exchange(self, solver, capacity_bytes, device)

    def shutdown(self):
        pass


class Group:
    def __init__(self, backend):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ["RANK"])
        self.world_size = int(os.environ["WORLD_SIZE"])
        self.local_rank = int(os.environ.get("LOCAL_RANK", self.rank))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        self.backend = backend
        self.comm = None
        if backend == "nccl":
            torch.cuda.set_device(self.local_rank)
            self.device = torch.device("cuda", self.local_rank)
        else:
            self.device = torch.device("cpu")
        # (device_id binds the communicator to this rank's GPU up front: no 'guessing device ID' and no lazy init in the first collective)
        kw = {"device_id": self.device} if backend == "nccl" else {}
        pg_backend = "gloo" if backend == "rccl" else backend       # native transport: torch only does the rendezvous, on the CPU
        # Two bounds.  A rank that never shows up must not hang the others for torch's default half hour: the RENDEZVOUS (an explicit
        # TCPStore that waits for every rank) is bounded by PHX_COMM_TIMEOUT_S, like csrc/comm.hip's.  The process group's own timeout —
        # which bounds every LATER collective of the group too — is the generous PHX_COLLECTIVE_TIMEOUT_S (default 1800 s): ranks are
        # legitimately out of step for minutes (rank 0 builds the library or a scene, runs the CPU baseline) and must not be
        # aborted for it.
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=collective_timeout_s())
        store = None
        try:
            store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), self.world_size, is_master=self.rank == 0,
                                  timeout=datetime.timedelta(seconds=comm_timeout_s()), wait_for_workers=True)
        except TypeError:                                   # a torch whose TCPStore has other arguments: env rendezvous under the short bound
            kw["timeout"] = datetime.timedelta(seconds=comm_timeout_s())
        if store is not None:
            kw["store"] = store
        try:
            dist.init_process_group(backend=pg_backend, rank=self.rank, world_size=self.world_size, **kw)
        except TypeError:                                   # a torch without the device_id argument
            kw.pop("device_id", None)
            dist.init_process_group(backend=pg_backend, rank=self.rank, world_size=self.world_size, **kw)
        self._flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        if backend == "rccl":
            # the library's own RCCL communicator (csrc/comm.hip): rank 0's id reaches everybody through the gloo group
            from . import api
            why = ""
            box = [None]
            if self.rank == 0:
                try:
                    box = [api.Comm.unique_id()]
                except Exception as e:             # no RCCL on this box
                    why = str(e)
            dist.broadcast_object_list(box, src=0)
            if box[0] is not None:
                try:
                    self.comm = api.Comm(box[0], self.rank, self.world_size, self.local_rank)
                except Exception as e:
                    why = str(e)
            # every rank must end up on the same transport
            ok = torch.tensor([1 if self.comm is not None else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                import sys
                self.comm = None
                self.backend = "gloo"
                if self.rank == 0:
                    sys.stderr.write("phyx_amd.dist: no native RCCL communicator (%s): the collectives of this run are staged through the host "
                                     "(gloo) — functional, not what a GPU node should measure\n" % (why or "a peer failed to create it"))

    def _sync(self):
        if self.backend == "nccl":
            self.torch.cuda.synchronize(self.device)

    def barrier(self):
        self.dist.barrier()
        self._sync()

    def step_barrier(self):
        """The per-step exchange of the island-sharded solve: one 4-byte all-reduce (done / error flag)."""
        return self.step_barrier_value(0)

    def step_barrier_value(self, value):
        """4-byte all-reduce (max) of this rank's status word; every rank gets the worst one."""
        self._flag.fill_(int(value))
        self.dist.all_reduce(self._flag, op=self.dist.ReduceOp.MAX)
        self._sync()
        return int(self._flag.item())

    def agree(self, status, nbytes):
        """(worst status, smallest, largest byte count) over all ranks: what step_sharded needs to know before the all-gather."""
        t = self.torch.tensor([int(status), int(nbytes), -int(nbytes)], dtype=self.torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self._sync()
        v = t.tolist()
        return int(v[0]), -int(v[2]), int(v[1])

    def info(self):
        """what this run's transport is, for bench.py's config"""
        from . import api
        v = api.Comm.rccl_version() if self.backend == "rccl" else 0
        return {"backend": self.backend, "ranks_seen": self.world_size, "rccl_version": v,
                "native_communicator": self.comm is not None}

    def stream_hook(self, stream_ptr):
        """hook(step, phase) for Solver.bench: the per-step exchange (a 4-byte all-reduce) lives ON THE SOLVER'S OWN STREAM.
        Phase 0 (step queued) starts it asynchronously behind the step; phase 1 (the next step's local preparation is
        queued, its sweeps are not) makes the stream wait for it — so the exchange overlaps the preparation, the sweeps
        of step s+1 start only after step s of every rank, and the host never blocks.
        (gloo has no streams: there phase 0 is the blocking all-reduce.)"""
        if self.backend == "rccl":                 # queued on the solver's stream by the library: the next step's kernels wait behind it
            return lambda step, phase: self.comm.barrier_async(stream_ptr) if phase == 0 else None
        if self.backend != "nccl":
            return lambda step, phase: self.step_barrier() if phase == 0 else None
        ext = self.torch.cuda.ExternalStream(stream_ptr, device=self.device)
        flag = self.torch.zeros(1, dtype=self.torch.int32, device=self.device)
        self.torch.cuda.synchronize(self.device)
        pending = []

        def hook(step, phase):
            with self.torch.cuda.stream(ext):
                if phase == 0:
                    pending.append(self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX, async_op=True))
                else:
                    while pending:
                        pending.pop(0).wait()          # the solver's stream waits for the exchange; the host does not
        self._hook_keep = (ext, flag, pending)
        return hook

    def all_gather_bytes(self, mine):
        """Host-side all-gather of equal-length uint8 arrays -> one array, rank-major (the gloo transport of Exchange)."""
        torch, dist = self.torch, self.dist
        mine = np.ascontiguousarray(mine, dtype=np.uint8)
        dev = self.device if self.backend == "nccl" else torch.device("cpu")      # (ProcessGroupNCCL takes device tensors only)
        parts = [torch.zeros(len(mine), dtype=torch.uint8, device=dev) for _ in range(self.world_size)]
        dist.all_gather(parts, torch.from_numpy(mine).to(dev))
        self._sync()
        return np.concatenate([p.cpu().numpy() for p in parts])

    def exchange(self, solver, capacity_bytes, device=None):
        return Exchange(self, solver, capacity_bytes, self.local_rank if device is None and self.backend in ("nccl", "rccl") else (device or 0))

    def _reduce(self, x, op):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=op)
        self._sync()
        return float(t.item())

    def reduce_max(self, x):
        return self._reduce(x, self.dist.ReduceOp.MAX)

    def reduce_sum(self, x):
        return self._reduce(x, self.dist.ReduceOp.SUM)

    def shutdown(self):
        self.comm = None
        try:
            self.dist.destroy_process_group()
        except Exception:
            pass


def collective_timeout_s():
    """seconds a collective of the torch process group may take before it is aborted (PHX_COLLECTIVE_TIMEOUT_S, default 1800)"""
    try:
        v = float(os.environ.get("PHX_COLLECTIVE_TIMEOUT_S", "0"))
    except ValueError:
        v = 0.0
    return v if v > 0 else 1800.0


def comm_timeout_s():
    """seconds a rank waits for its peers before it gives up (PHX_COMM_TIMEOUT_S, default 120; csrc/comm.hip reads the same)"""
    try:
        v = float(os.environ.get("PHX_COMM_TIMEOUT_S", "0"))
    except ValueError:
        v = 0.0
    return v if v > 0 else 120.0


def watchdog(seconds, on_expiry):
    """Arms a timer thread that calls on_expiry() and then leaves the process with status 3 — for a rank that may block inside a
    collective some peer never joins (ctypes and torch release the GIL there).  Returns the timer: .cancel() it when done."""
    import threading

    def fire():
        try:
            on_expiry()
        finally:
            os._exit(3)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def self_launch(n, argv=None, timeout_s=None):
    """`python bench.py --gpus N` started WITHOUT a launcher (WORLD_SIZE unset): spawn the N ranks ourselves — one process per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, the same command line — and wait for them.  Rank 0
    prints the one JSON line (the children inherit stdout / stderr).  Returns the exit status to leave with.
    A rank that dies takes the others down with it (they would wait for it in the next collective), and the whole run is bounded
    by `timeout_s` (PHX_LAUNCH_TIMEOUT_S, default 1800): the launcher never hangs."""
    import socket
    import subprocess
    import sys
    import time
    argv = list(sys.argv if argv is None else argv)
    if timeout_s is None:
        try:
            timeout_s = float(os.environ.get("PHX_LAUNCH_TIMEOUT_S", "0")) or 1800.0
        except ValueError:
            timeout_s = 1800.0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable] + argv, env=env))
    rc, t0 = 0, time.time()
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            rc = max(rc, abs(code))
        if live and (rc != 0 or time.time() - t0 > timeout_s):
            if rc == 0:
                rc = 124
                sys.stderr.write("phyx_amd.dist.self_launch: %d rank(s) still running after %.0f s: terminating them\n" % (len(live), timeout_s))
            for p in live:                         # (exactly the processes started above)
                p.terminate()
            deadline = time.time() + 10.0
            for p in live:
                try:
                    p.wait(max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            live = []
    return rc


def init(n_gpus, backend="rccl", force=False):
    """Returns the process group wrapper; a plain single-process object when WORLD_SIZE is 1/unset."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws == 1 and not force:
        if n_gpus != 1:
            raise SystemExit("--gpus %d needs one process per GPU (bench.py launches them itself when WORLD_SIZE is unset)" % n_gpus)
        return Single()
    if ws != n_gpus and not force:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (ws, n_gpus))
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    return Group(backend)


def shard_columns(total_columns, rank, world_size):
    """Contiguous slab of stack columns (= islands) owned by `rank`: (first_column, column_count)."""
    base, extra = divmod(total_columns, world_size)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


# ---- ownership sharding: one world per rank, each holding the islands of one x-slab ---------------------------------------
# BASELINE config 3 read literally ("islands shard naturally across GPUs … RCCL only for the per-step barrier"): instead of a
# replica of the whole world plus an all-gather of everybody's results (Exchange above), a rank SIMULATES only the bodies whose
# centre lies in its slab of the x axis — the broadphase's sweep axis, so slabs are natural — plus every static body.  As long
# as no dynamic body reaches across a slab boundary, no contact, hence no island, spans two ranks: the ranks' worlds are
# independent sub-problems of the reference's island loop (ref: Solver.cpp:86-91), the per-step collective is a 4-byte
# all-reduce, and nothing else ever crosses xGMI.  The guard (SlabWorld.check) watches that condition on the device
# (phx_world_x_extent) and every rank learns of a violation at the same step.  The answer to a violation is a RE-SLAB
# (SlabWorld.reslab): the ranks all-gather their worlds' states — bodies, manifolds with their contact points, joints with their
# warm-start impulses, i.e. what phx_world_set_state restores — cut the x axis anew in the gaps no dynamic body's AABB covers
# (slab_cuts: bodies that touch, or share a manifold, always stay together), and every rank restores the sub-world of its new slab.
# Islands may so wander, merge and topple across the old boundaries; the hand-over happens between two steps, before the broadphase
# could have missed a pair (the guard looks at the AABBs the NEXT step's broadphase will sweep).  A world that has merged into one
# island leaves no gap to cut in: it ends up on one rank, and replica mode (Exchange above) is the mode for it.
# gather_bodies() assembles the full world on demand.
#
# A slab world is bit-exact against the oracle of ITS OWN slab scene (tests/test_world_gpu.py), not against the unsharded world:
# colouring priorities hash the contact-point index, which is local to a world, so the two sweep the same islands in different
# (equally legal) Gauss-Seidel orders — like the reference's own solve modes differ among themselves.

def slab_partition(scene, nranks):
    """Cuts a scene into `nranks` x-slabs with (nearly) equal numbers of dynamic bodies.  Returns a list of
    (sub_scene, global_index, (lo, hi)): the rank's bodies in their original relative order (static bodies of the whole scene
    included in every slab), their indices in the full scene, and the slab's open x-interval (−inf / +inf at the ends)."""
    px = np.asarray(scene["px"], dtype=np.float64)
    static = np.asarray(scene["static"], dtype=bool)
    dyn = np.flatnonzero(~static)
    order = dyn[np.argsort(px[dyn], kind="stable")]
    cuts = [len(order) * r // nranks for r in range(nranks + 1)]
    for c in range(1, nranks):                     # never cut between bodies with the same centre x (a column of a stack)
        cuts[c] = max(cuts[c], cuts[c - 1])
        while 0 < cuts[c] < len(order) and px[order[cuts[c]]] == px[order[cuts[c] - 1]]:
            cuts[c] += 1
    out = []
    for r in range(nranks):
        mine = order[cuts[r]:cuts[r + 1]]
        # a slab boundary sits half-way between the neighbouring bodies' centres; equal centres must not be split
        lo = -np.inf if cuts[r] == 0 or not len(mine) else 0.5 * (px[order[cuts[r] - 1]] + px[mine[0]])
        hi = np.inf if cuts[r + 1] >= len(order) or not len(mine) else 0.5 * (px[mine[-1]] + px[order[cuts[r + 1]]])
        keep = np.sort(np.concatenate([np.flatnonzero(static), mine]))
        sub = {k: np.asarray(v)[keep] for k, v in scene.items() if hasattr(v, "__len__") and len(v) == len(px)}
        out.append((sub, keep, (float(lo), float(hi))))
    return out


def slab_cuts(lo, hi, nranks, margin=0.0):
    """Island-safe slabs for dynamic bodies whose AABBs span [lo[i], hi[i]] on the x axis.  Bodies whose (margin-widened) intervals
    overlap form a block that is never split; the nranks - 1 cuts sit in the middle of the gaps between blocks, chosen so that
    the slabs hold nearly equal numbers of bodies.  Returns (owner, bounds): the rank of every body and the ranks' open x-intervals
    (a rank may end up with no body at all if there are fewer blocks than ranks)."""
    lo = np.asarray(lo, dtype=np.float64); hi = np.asarray(hi, dtype=np.float64)
    n = len(lo)
    owner = np.zeros(n, dtype=np.int64)
    if n == 0:
        return owner, [(-np.inf, np.inf)] * nranks
    order = np.argsort(lo, kind="stable")
    slo, shi = lo[order] - margin, hi[order] + margin
    reach = np.maximum.accumulate(shi)
    starts = np.flatnonzero(np.concatenate([[True], slo[1:] > reach[:-1]]))          # first sorted position of every block
    ends = np.concatenate([starts[1:], [n]])                                           # one past its last
    cut_at = [0]                                                                       # indices into `starts`: block where each slab begins
    for r in range(1, nranks):
        k = int(np.argmin(np.abs(ends - n * r / nranks)))                              # the block end nearest to the ideal count
        cut_at.append(min(max(k + 1, cut_at[-1]), len(starts)))
    cut_at.append(len(starts))
    true_hi = np.maximum.accumulate(hi[order])
    bounds = []
    for r in range(nranks):
        b0, b1 = cut_at[r], cut_at[r + 1]
        if b0 >= b1:
            bounds.append((np.inf, np.inf) if b0 >= len(starts) else (np.nan, np.nan))
            continue
        first, last = starts[b0], ends[b1 - 1]
        owner[order[first:last]] = r
        left = -np.inf if first == 0 else 0.5 * (float(true_hi[first - 1]) + float(lo[order[first]]))
        right = np.inf if last == n else 0.5 * (float(true_hi[last - 1]) + float(lo[order[last]]))
        bounds.append((left, right))
    # a rank without bodies gets an empty interval at its neighbour's edge (nothing lives there, nothing can violate it)
    bounds = [(b if not (np.isnan(b[0]) or b[0] == np.inf) else (np.inf, np.inf)) for b in bounds]
    return owner, bounds


def _blob(*arrays):
    """numpy arrays -> one uint8 array: int64 count + int64 byte sizes + the arrays' bytes (8-byte aligned)"""
    parts = [np.ascontiguousarray(a).view(np.uint8).reshape(-1) for a in arrays]
    head = np.array([len(parts)] + [len(q) for q in parts], dtype=np.int64).view(np.uint8)
    out = [head]
    for q in parts:
        out.append(q)
        out.append(np.zeros((-len(q)) % 8, dtype=np.uint8))
    return np.concatenate(out)


def _unblob(blob, dtypes):
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    k = int(blob[:8].view(np.int64)[0])
    sizes = blob[8:8 + 8 * k].view(np.int64)
    at, out = 8 + 8 * k, []
    for size, dt in zip(sizes, dtypes):
        out.append(blob[at:at + int(size)].view(dt).copy())
        at += int(size) + (-int(size)) % 8
    return out


def all_gather_blobs(group, mine):
    """all-gather of byte strings of different lengths -> list of uint8 arrays, one per rank"""
    mine = np.ascontiguousarray(mine, dtype=np.uint8)
    if group.world_size == 1:
        return [mine]
    longest = int(group.reduce_max(len(mine)))
    buf = np.zeros(8 + longest + (-longest) % 8, dtype=np.uint8)
    buf[:8] = np.array([len(mine)], dtype=np.int64).view(np.uint8)
    buf[8:8 + len(mine)] = mine
    rows = group.all_gather_bytes(buf).reshape(group.world_size, -1)
    return [rows[r, 8:8 + int(rows[r, :8].view(np.int64)[0])] for r in range(group.world_size)]


class SlabWorld:
    """One rank of an ownership-sharded world (see the comment above)."""

    def __init__(self, group, scene, device=0, gravity=0.0, check_every=1, auto_reslab=True, reslab_every=0, margin=1.0):
        """`auto_reslab`: a violated guard re-slabs the world (reslab()) instead of raising; `reslab_every` > 0 re-slabs every
        so many steps whatever the guard says (load balance; tests); `margin`: how far beyond a body's AABB a cut keeps away."""
        import phyx_amd
        self.group = group
        sub, self.global_index, self.bounds = slab_partition(scene, group.world_size)[group.rank]
        self.scene_size = len(scene["px"])
        self.device, self.gravity = device, gravity
        self.world = phyx_amd.World(device, gravity=gravity)
        self.world.add_scene(sub)
        self.check_every, self.steps = check_every, 0
        self.auto_reslab, self.reslab_every, self.margin, self.reslabs = auto_reslab, reslab_every, margin, 0

    def step(self, dt, configuration):
        """World::Update of this rank's slab, then the per-step barrier (a 4-byte all-reduce that carries the guard's verdict);
        a violated guard — on any rank — is answered by a re-slab on every rank, before the next step."""
        self.world.Update(dt, configuration)
        self.steps += 1
        bad = 0
        if self.steps % self.check_every == 0:
            bad = 0 if self.inside() else 1
        flag = self.group.step_barrier_value(bad) if hasattr(self.group, "step_barrier_value") else bad
        if flag and not self.auto_reslab:
            raise RuntimeError("ownership-sharded world: a body reached the boundary of its slab (rank %d, slab %s); the islands of two "
                               "ranks may now touch — re-slab the world (reslab) before going on" % (self.group.rank, self.bounds))
        if flag or (self.reslab_every > 0 and self.steps % self.reslab_every == 0):
            self.reslab()

    # ---- re-slab: hand the worlds' states over to new owners -----------------------------------------------------------------
    _STATE_DTYPES = None

    def reslab_pack(self):
        """This rank's share of a re-slab: its world's state with body ids in the full scene's numbering, as one byte string."""
        import phyx_amd
        bodies, manifolds, cps, joints = self.world.state()
        gi = np.asarray(self.global_index, dtype=np.int64)
        m = manifolds.copy(); j = joints.copy()
        if len(m):
            m["body1"] = gi[m["body1"]]; m["body2"] = gi[m["body2"]]
        if len(j):
            j["body1"] = gi[j["body1"]]; j["body2"] = gi[j["body2"]]
        return _blob(gi, bodies, m, cps, j)

    def reslab_apply(self, blobs):
        """Every rank's reslab_pack() -> this rank's new slab: the union world is assembled (bodies in scene order; manifolds, contact
        points and joints rank after rank), cut anew (slab_cuts on the dynamic bodies' AABBs, bodies sharing a manifold kept
        together), and the part of it that lives in this rank's new slab is restored into a fresh World."""
        import phyx_amd
        dts = (np.int64, phyx_amd.rigid_body_dtype, phyx_amd.manifold_dtype, phyx_amd.contact_point_dtype, phyx_amd.contact_joint_dtype)
        full = np.zeros(self.scene_size, dtype=phyx_amd.rigid_body_dtype)
        seen = np.zeros(self.scene_size, dtype=bool)
        ms, cs, js = [], [], []
        m_off = j_off = 0
        for blob in blobs:
            gi, b, m, c, j = _unblob(blob, dts)
            full[gi] = b; seen[gi] = True
            m = m.copy(); c = c.copy(); j = j.copy()
            m["point_index"] = 2 * (np.arange(len(m), dtype=np.int32) + m_off)
            j["contact_point_index"] += 2 * m_off
            ms.append(m); cs.append(c); js.append(j)
            m_off += len(m); j_off += len(j)
        if not seen.all():
            raise RuntimeError("re-slab: %d bodies of the scene are on no rank" % int((~seen).sum()))
        M = np.concatenate(ms) if ms else np.zeros(0, dtype=phyx_amd.manifold_dtype)
        Cp = np.concatenate(cs) if cs else np.zeros(0, dtype=phyx_amd.contact_point_dtype)
        J = np.concatenate(js) if js else np.zeros(0, dtype=phyx_amd.contact_joint_dtype)
        static = (full["inv_mass"] == 0) & (full["inv_inertia"] == 0)
        dyn = np.flatnonzero(~static)
        lo = full["aabb_min"]["x"].astype(np.float64); hi = full["aabb_max"]["x"].astype(np.float64)
        if len(M):                                           # two dynamic bodies that share a manifold cover each other's interval
            both = ~static[M["body1"]] & ~static[M["body2"]]
            for _ in range(2):
                a, b = M["body1"][both], M["body2"][both]
                l = np.minimum(lo[a], lo[b]); h = np.maximum(hi[a], hi[b])
                np.minimum.at(lo, a, l); np.minimum.at(lo, b, l); np.maximum.at(hi, a, h); np.maximum.at(hi, b, h)
        owner_dyn, bounds = slab_cuts(lo[dyn], hi[dyn], self.group.world_size, self.margin)
        owner = np.full(self.scene_size, -1, dtype=np.int64)
        owner[dyn] = owner_dyn
        me = self.group.rank
        keep = np.flatnonzero(static | (owner == me))
        local_of = np.full(self.scene_size, -1, dtype=np.int64)
        local_of[keep] = np.arange(len(keep))
        if len(M):
            m_owner = np.where(static[M["body1"]], owner[M["body2"]], owner[M["body1"]])
            both = ~static[M["body1"]] & ~static[M["body2"]]
            if np.any(owner[M["body1"]][both] != owner[M["body2"]][both]):
                raise RuntimeError("re-slab: a manifold spans two slabs")
            sel = np.flatnonzero(m_owner == me)
        else:
            sel = np.zeros(0, dtype=np.int64)
        new_m_of = np.full(len(M), -1, dtype=np.int64)
        new_m_of[sel] = np.arange(len(sel))
        m = M[sel].copy()
        if len(m):
            m["body1"] = local_of[m["body1"]]; m["body2"] = local_of[m["body2"]]
            m["point_index"] = 2 * np.arange(len(m), dtype=np.int32)
        c = Cp.reshape(-1, 2)[sel].reshape(-1).copy() if len(sel) else np.zeros(0, dtype=phyx_amd.contact_point_dtype)
        jsel = np.flatnonzero(new_m_of[J["contact_point_index"] // 2] >= 0) if len(J) else np.zeros(0, dtype=np.int64)
        j = J[jsel].copy()
        if len(j):                                           # (slots no joint points at keep their bytes: the step never reads them)
            j["contact_point_index"] = (2 * new_m_of[j["contact_point_index"] // 2] + j["contact_point_index"] % 2).astype(np.int32)
            j["body1"] = local_of[j["body1"]]; j["body2"] = local_of[j["body2"]]
            c["solver_index"][j["contact_point_index"]] = np.arange(len(j), dtype=np.int32)
        b = full[keep].copy()
        b["index"] = np.arange(len(keep), dtype=np.uint32)
        world = phyx_amd.World(self.device, gravity=self.gravity)
        world.set_state(b, m, c, j)
        self.world, self.global_index, self.bounds = world, keep, (float(bounds[me][0]), float(bounds[me][1]))
        self.reslabs += 1

    def reslab_intervals(self):
        """This rank's share of the re-slab's FIRST phase: {scene index, x-interval} of its dynamic bodies — the intervals reslab_apply
        computes from the union world (two bodies that share a manifold cover each other's interval: a manifold never spans two
        ranks, so every rank widens its own), 24 bytes per body instead of the world's whole state."""
        bodies, manifolds, _, _ = self.world.state()
        static = (bodies["inv_mass"] == 0) & (bodies["inv_inertia"] == 0)
        lo = bodies["aabb_min"]["x"].astype(np.float64); hi = bodies["aabb_max"]["x"].astype(np.float64)
        if len(manifolds):
            both = ~static[manifolds["body1"]] & ~static[manifolds["body2"]]
            for _ in range(2):
                a, b = manifolds["body1"][both], manifolds["body2"][both]
                l = np.minimum(lo[a], lo[b]); h = np.maximum(hi[a], hi[b])
                np.minimum.at(lo, a, l); np.minimum.at(lo, b, l); np.maximum.at(hi, a, h); np.maximum.at(hi, b, h)
        dyn = np.flatnonzero(~static)
        return _blob(np.asarray(self.global_index, dtype=np.int64)[dyn], lo[dyn], hi[dyn])

    def reslab_plan(self, interval_blobs):
        """Every rank's reslab_intervals() -> (the new owner of every dynamic body of the scene in scene order, their scene indices, the
        ranks' new bounds): the same slab_cuts on the same intervals as reslab_apply, computed without the worlds' states."""
        gis, los, his = [], [], []
        for blob in interval_blobs:
            gi, lo, hi = _unblob(blob, (np.int64, np.float64, np.float64))
            gis.append(gi); los.append(lo); his.append(hi)
        gi = np.concatenate(gis); lo = np.concatenate(los); hi = np.concatenate(his)
        order = np.argsort(gi, kind="stable")
        gi, lo, hi = gi[order], lo[order], hi[order]
        owner, bounds = slab_cuts(lo, hi, self.group.world_size, self.margin)
        return owner, gi, bounds

    def reslab(self):
        """Collective: every rank of the group must call it at the same step (step() does, on the all-reduced verdict).
        A caller of the library's phx_world_reslab (csrc/reslab.hip) since round 6: the planning is the library's host code, the
        collectives run on device buffers through the group's native RCCL communicator — or, where the group has none (gloo: several
        ranks on one GPU), through this group's own all-gather / all-reduce as the library's two host callbacks.  reslab_python() below is
        the round-5 statement of the same hand-over in numpy; the tests hold the two against each other."""
        if getattr(self, "python_reslab", False):
            return self.reslab_python()
        g = self.group
        native = getattr(g, "comm", None)
        kw = {"comm": native} if native is not None else ({"all_gather": g.all_gather_bytes, "all_reduce_max": lambda v: int(g.reduce_max(v))} if g.world_size > 1 else {})
        world_before = self.world
        moved, gi, bounds = self.world.reslab(self.global_index, self.scene_size, self.bounds, margin=self.margin, rank=g.rank, size=g.world_size, **kw)
        self.global_index, self.bounds = gi, bounds
        self.reslabs += 1
        if not moved:
            self.reslabs_in_place = getattr(self, "reslabs_in_place", 0) + 1
        assert self.world is world_before

    def reslab_python(self):
        """The same hand-over stated in numpy over host-staged collectives (round 5).
        Two phases.  First the ranks all-gather only their dynamic bodies' x-intervals (24 bytes per body) and cut the axis anew; if no
        body changes its owner — the usual case of a guard hit: a pile leaned over its old cut but the gaps are where they were —
        every rank KEEPS its World (allocations, cached schedule, broadphase splitters and all) and only takes its new bounds.  Only
        when some body does move are the worlds' states gathered and restored (reslab_apply): that costs O(whole world) per rank
        (the blobs, their padding to the longest one and the union arrays: several hundred MB at 1e6 bodies), a price paid per
        migration, not per guard hit."""
        owner, gi, bounds = self.reslab_plan(all_gather_blobs(self.group, self.reslab_intervals()))
        me = self.group.rank
        mine_now = np.sort(np.asarray(self.global_index, dtype=np.int64)[np.isin(np.asarray(self.global_index, dtype=np.int64), gi)])
        mine_new = np.sort(gi[owner == me])
        moved = 0 if np.array_equal(mine_now, mine_new) else 1
        moved = int(self.group.reduce_max(moved)) if self.group.world_size > 1 else moved
        if not moved:
            self.bounds = (float(bounds[me][0]), float(bounds[me][1]))
            self.reslabs += 1
            self.reslabs_in_place = getattr(self, "reslabs_in_place", 0) + 1
            return
        self.reslab_apply(all_gather_blobs(self.group, self.reslab_pack()))

    def inside(self):
        """True iff every dynamic body's AABB lies strictly inside this rank's slab: then no AABB of this rank overlaps one of
        another rank, the broadphase of the whole world would find no pair between them, and no island spans two ranks."""
        if self.bounds[0] == np.inf:                        # (a re-slab left this rank without dynamic bodies)
            return True
        lo, hi = self.world.x_extent()
        return lo > self.bounds[0] and hi < self.bounds[1]

    def check(self):
        flag = 0 if self.inside() else 1
        flag = self.group.step_barrier_value(flag) if hasattr(self.group, "step_barrier_value") else flag
        return flag == 0

    def gather_bodies(self):
        """The whole world's body records, in the full scene's index order, on every rank (static bodies from rank 0)."""
        import phyx_amd
        mine = self.world.bodies
        full = np.zeros(self.scene_size, dtype=phyx_amd.rigid_body_dtype)
        if self.group.world_size == 1:
            full[self.global_index] = mine
            return full
        n_max = int(self.group.reduce_max(len(mine)))
        buf = np.zeros(n_max * mine.dtype.itemsize + 8 * n_max + 8, dtype=np.uint8)
        buf[:8] = np.frombuffer(np.int64(len(mine)).tobytes(), dtype=np.uint8)
        buf[8:8 + 8 * len(mine)] = np.frombuffer(self.global_index.astype(np.int64).tobytes(), dtype=np.uint8)
        buf[8 + 8 * n_max:8 + 8 * n_max + mine.nbytes] = np.frombuffer(mine.tobytes(), dtype=np.uint8)
        allb = self.group.all_gather_bytes(buf).reshape(self.group.world_size, -1)
        for r in range(self.group.world_size - 1, -1, -1):           # (rank 0 last: its copy of the static bodies wins)
            n = int(np.frombuffer(allb[r, :8].tobytes(), dtype=np.int64)[0])
            idx = np.frombuffer(allb[r, 8:8 + 8 * n].tobytes(), dtype=np.int64)
            rec = np.frombuffer(allb[r, 8 + 8 * n_max:8 + 8 * n_max + n * mine.dtype.itemsize].tobytes(), dtype=phyx_amd.rigid_body_dtype)
            full[idx] = rec
        full["index"] = np.arange(self.scene_size, dtype=np.uint32)
        return full
