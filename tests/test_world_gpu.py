"""World::Update parity (MI355X): the device-resident World (every stage of the step a HIP kernel) against the oracle
World, step by step.  The oracle's solver is driven in the
device's colour order (see test_solver_gpu.py); with that, every byte of every body, manifold, contact
point and joint must agree after every step."""
import numpy as np
import pytest

import phyx_amd
from phyx_amd import scenes, Configuration
from helpers import oracle_world

pytestmark = pytest.mark.gpu


def _lockstep(oracle, scene, steps, cfg, check_every=1):
    pw = phyx_amd.World(0, gravity=-200.0)
    pw.add_scene(scene)
    ow = oracle_world(scene)
    assert pw.bodies.tobytes() == ow.bodies().tobytes()                    # AddBody / RigidBody ctor
    for step in range(steps):
        pw.Update(1.0 / 60.0, cfg)
        ow.pre_solve(1.0 / 60.0)
        order, offs = pw.solver.schedule()
        groups, _ = pw.solver.groups()
        b, cp, j = ow.bodies(), ow.contact_points(), ow.joints()           # live views into the oracle world
        assert len(order) == len(j)
        oracle.solver_solve_grouped(b, cp, j, order, offs, groups, cfg.contactIterationsCount, cfg.penetrationIterationsCount,
                                    oracle.STAG_COLOUR_SYNC)
        ow.integrate_position(1.0 / 60.0)
        if step % check_every == 0 or step == steps - 1:
            assert pw.counts() == (len(ow.bodies()), len(ow.manifolds()), len(ow.contact_points()), len(ow.joints())), "step %d" % step
            assert pw.manifolds.tobytes() == ow.manifolds().tobytes(), "manifolds differ at step %d" % step
            assert pw.contactJoints.tobytes() == ow.joints().tobytes(), "joints differ at step %d" % step
            assert pw.bodies.tobytes() == ow.bodies().tobytes(), "bodies differ at step %d" % step
            m = ow.manifolds()
            live = np.concatenate([np.arange(int(x["point_index"]), int(x["point_index"]) + int(x["point_count"])) for x in m] + [np.zeros(0, dtype=np.int64)]).astype(np.int64)
            assert pw.contactPoints[live].tobytes() == ow.contact_points()[live].tobytes(), "contact points differ at step %d" % step
    return pw, ow


@pytest.mark.parametrize("name,steps", [("stack", 12), ("tilted", 60), ("falling", 50), ("clique", 5), ("wall", 16)])
@pytest.mark.parametrize("island_mode", [0, 3])
def test_world_lockstep_bit_exact(oracle, built_lib, name, steps, island_mode):
    scene = {"stack": lambda: scenes.stack(6, 40), "tilted": lambda: scenes.tilted(80),
             "falling": lambda: scenes.falling(500, width=80.0, ymax=300.0),
             "clique": lambda: scenes.clique(90),                           # > 64 colours: host-builder fallback inside a World
             "wall": lambda: scenes.wall(40, 36)}[name]()                   # one island that grows past 1024 joints: partitioned, three parts
    cfg = Configuration(phyx_amd.SOLVE_SCALAR, island_mode, 15, 15)
    pw, ow = _lockstep(oracle, scene, steps, cfg)
    assert len(ow.joints()) > 0
    if name == "wall":
        ki, parts, _ = pw.solver.partition()
        assert ki > 0 and parts == 2 * 3 + 1


def test_late_manifold_pack_is_repeated_bit_exactly(oracle, built_lib):
    """PackManifolds' dead-manifold count is not waited for when the previous step found none: the joint match is queued on the bet
    that nothing dies and both counts come back in one round trip; if a manifold did die the pack runs then and the match is
    repeated under a new epoch (csrc/world.hip refresh_contact_joints).  A falling pile loses that bet now and then: the steps in
    which it does must still equal the oracle's, byte for byte."""
    scene = scenes.falling(400, width=70.0, ymax=260.0)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE_SLOPPY, 10, 10)
    pw, ow = _lockstep(oracle, scene, 70, cfg)
    c = pw.debug_counters()
    assert c["deferred_packs"] > 0 and c["deferred_pack_retries"] > 0, c


@pytest.mark.parametrize("scene", [0, 2, 3, 4, 5, 6, 7])
def test_reference_demo_scenes_lockstep(oracle, built_lib, scene):
    """Headless, scaled-down versions of the reference's demo scenes (ref: main.cpp:97-227, minus the 'Wall' that overflows the
    reference's own hash set) in lockstep with the oracle World: pyramids (wide boxes on narrow ones), tapered stacks, shelves
    pinned with invMass = 0 only (they rotate under load), a tilted static plank, static splitters between islands."""
    sc = scenes.reference(scene, boxes=360)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE_SLOPPY if scene % 2 else phyx_amd.ISLAND_SINGLE, 15, 15)
    pw, ow = _lockstep(oracle, sc, 30, cfg, check_every=3)
    assert len(ow.joints()) > 0
    if "pinned" in sc and sc["pinned"].any():
        b = pw.bodies
        assert (b["inv_mass"][sc["pinned"]] == 0).all() and (b["inv_inertia"][sc["pinned"]] > 0).all()


def test_differential_fuzz_random_worlds(built_lib):
    """tools/fuzz.py: random worlds (random sizes, angles, overlaps, static shelves, random island mode and iteration counts)
    in lockstep with the oracle, every byte compared after every step.  40 seeds here; 46 000 (and 870 of the --big kind) were run for round 1, 30 000 + 750 on round 2's code."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz.py"), "50000", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 diverged" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


class _LocalRanks:
    """k sharded Worlds on ONE GPU: the all-gather of the island-sharded exchange emulated with device-to-device copies
    (every rank's segment into every rank's recv buffer at offset rank * segment_bytes) — what RCCL does on a node."""

    def __init__(self, scene, k, capacity, gravity=-200.0):
        self.k = k
        self.worlds, self.send, self.recv = [], [], []
        for r in range(k):
            w = phyx_amd.World(0, gravity=gravity)
            w.add_scene(scene)
            w.set_shard(r, k)
            snd, rcv = phyx_amd.DeviceBuffer(capacity, 0), phyx_amd.DeviceBuffer(capacity * k, 0)
            w.solver.set_exchange_buffers(snd.ptr.value, rcv.ptr.value, capacity)
            self.worlds.append(w); self.send.append(snd); self.recv.append(rcv)

    def step(self, dt, cfg):
        segs = [w.StepBegin(dt, cfg) for w in self.worlds]
        assert len(set(segs)) == 1, "every rank must derive the same segment size: %r" % (segs,)
        for w in self.worlds:
            w.sync()
        for dst in range(self.k):
            for src in range(self.k):      # ordered on the destination world's stream, in front of its unpack
                self.recv[dst].copy_from(self.send[src], segs[0], dst_offset=src * segs[0], stream=self.worlds[dst].stream_ptr())
        for w in self.worlds:
            w.StepEnd(dt)
        return segs[0]


def _same_world(a, b, what):
    assert a.counts() == b.counts(), what
    assert a.bodies.tobytes() == b.bodies.tobytes(), "bodies differ: " + what
    assert a.contactJoints.tobytes() == b.contactJoints.tobytes(), "joints differ: " + what
    assert a.manifolds.tobytes() == b.manifolds.tobytes(), "manifolds differ: " + what


@pytest.mark.parametrize("name,k,steps,mode", [("stack", 3, 14, phyx_amd.ISLAND_MULTIPLE), ("falling", 3, 40, phyx_amd.ISLAND_MULTIPLE_SLOPPY),
                                               ("stack", 8, 10, phyx_amd.ISLAND_MULTIPLE), ("pile", 2, 12, phyx_amd.ISLAND_MULTIPLE)])
def test_sharded_worlds_stay_in_lockstep_with_the_unsharded_world(oracle, built_lib, name, k, steps, mode):
    """BASELINE config 3: every rank steps a replica of the world, solves the schedule groups
    g % k == rank, and the ranks' results are all-gathered before IntegratePosition (the counterpart of the reference
    merging its islands back, ref: Solver.cpp:86-91, 482-494).  k sharded Worlds on one GPU with the all-gather emulated
    by device copies must match the unsharded World byte for byte after EVERY step — bodies, joints (warm-start impulses)
    and manifolds — and every replica must equal every other.  'pile' has an island too big for a workgroup: the
    HBM group is owned by one rank and exchanged like any other."""
    scene = {"stack": lambda: scenes.stack(64, 30), "falling": lambda: scenes.falling(4000, width=5000.0, ymax=260.0),
             "pile": lambda: scenes.falling(1500, width=60.0, ymax=700.0)}[name]()
    cfg = Configuration(phyx_amd.SOLVE_SCALAR, mode, 15, 15)
    full = phyx_amd.World(0, gravity=-200.0)
    full.add_scene(scene)
    from phyx_amd.dist import Exchange
    ranks = _LocalRanks(scene, k, Exchange.capacity_for(len(scene["px"]), 8 * len(scene["px"])))
    saw_groups = saw_hbm = 0
    for step in range(steps):
        full.Update(1.0 / 60.0, cfg)
        ranks.step(1.0 / 60.0, cfg)
        for r, w in enumerate(ranks.worlds):
            _same_world(w, full, "rank %d of %d at step %d" % (r, k, step))
        groups, lds = full.solver.groups()
        saw_groups = max(saw_groups, len(groups) - 1)
        saw_hbm |= int(len(groups) - 1 > lds)
    assert saw_groups >= (8 if name == "stack" else 2), saw_groups
    assert name != "pile" or saw_hbm, "the pile scene should have produced an island that only the HBM path can take"
    for w in ranks.worlds:
        assert w.solver.exchange_status() == 0


def test_sharded_world_at_config3_size_matches_the_oracle(oracle, built_lib):
    """BASELINE config 3 at full size: stack(1000,200) = 200 001 bodies, Multiple island mode, 8 shards (emulated on one
    GPU).  Two steps in lockstep with the oracle World driven in the device's group / colour order: every rank's replica
    must equal the oracle byte for byte after each step."""
    scene = scenes.stack(1000, 200)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE, 20, 20)
    from phyx_amd.dist import Exchange
    k = 8
    ranks = _LocalRanks(scene, k, Exchange.capacity_for(200001, 450000) // 1024 * 256)
    ow = oracle_world(scene)
    for step in range(2):
        seg = ranks.step(1.0 / 60.0, cfg)
        assert seg * k < 4 * Exchange.capacity_for(200001, 450000)
        ow.pre_solve(1.0 / 60.0)
        w0 = ranks.worlds[0]
        order, offs = w0.solver.schedule()
        groups, _ = w0.solver.groups()
        b, cp, j = ow.bodies(), ow.contact_points(), ow.joints()
        assert len(order) == len(j)
        oracle.solver_solve_grouped(b, cp, j, order, offs, groups, 20, 20, oracle.STAG_COLOUR_SYNC)
        ow.integrate_position(1.0 / 60.0)
        for r in (0, 3, 7):
            w = ranks.worlds[r]
            assert w.bodies.tobytes() == ow.bodies().tobytes(), "rank %d bodies differ from the oracle at step %d" % (r, step)
            assert w.contactJoints.tobytes() == ow.joints().tobytes(), "rank %d joints differ from the oracle at step %d" % (r, step)
    st = ranks.worlds[0].solver.stats()
    assert st.island_count >= 900 and st.lds_islands >= 900
    assert all(w.solver.exchange_status() == 0 for w in ranks.worlds)


def test_slab_worlds_at_config3_size_match_their_oracles(oracle, built_lib):
    """BASELINE config 3 in SLAB mode — what `bench.py --gpus 8` runs by default — at full size: stack(1000,200) cut into 8 x-slabs of
    125 columns (+ the ground), one World per slab (here all on GPU 0), five steps.  Every slab world equals the oracle world of
    its own sub-scene byte for byte after every step, the guard (no body near its slab's boundary) holds, and against the UNSHARDED
    200k world the union stays inside the stated tolerance (another legal Gauss-Seidel order: a slab numbers its contact points
    locally): manifold counts within 3 %, max |position difference| < 0.5 after five steps."""
    import types
    from phyx_amd import dist as pdist
    scene = scenes.stack(1000, 200)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE, 20, 20)
    k = 8
    parts = pdist.slab_partition(scene, k)
    assert sum(len(idx) - 1 for _, idx, _ in parts) == 200000
    slabs, oracles = [], []
    for r in range(k):
        g = types.SimpleNamespace(rank=r, world_size=k, step_barrier_value=lambda v: v)
        slabs.append(pdist.SlabWorld(g, scene, device=0, gravity=-200.0))
        oracles.append(oracle_world(parts[r][0]))
    full = phyx_amd.World(0, gravity=-200.0)
    full.add_scene(scene)
    for step in range(5):
        full.Update(1.0 / 60.0, cfg)
        for r in range(k):
            slabs[r].step(1.0 / 60.0, cfg)                                  # (raises if the guard trips)
            w, ow = slabs[r].world, oracles[r]
            ow.pre_solve(1.0 / 60.0)
            order, offs = w.solver.schedule()
            groups, _ = w.solver.groups()
            oracle.solver_solve_grouped(ow.bodies(), ow.contact_points(), ow.joints(), order, offs, groups, 20, 20, oracle.STAG_COLOUR_SYNC)
            ow.integrate_position(1.0 / 60.0)
            assert w.counts() == (len(ow.bodies()), len(ow.manifolds()), len(ow.contact_points()), len(ow.joints())), "slab %d step %d" % (r, step)
            assert w.bodies.tobytes() == ow.bodies().tobytes(), "slab %d bodies differ from its oracle at step %d" % (r, step)
            assert w.contactJoints.tobytes() == ow.joints().tobytes(), "slab %d joints differ from its oracle at step %d" % (r, step)
    st = slabs[0].world.solver.stats()
    assert st.lds_islands >= 100 and all(s.check() for s in slabs)
    union = np.zeros(len(scene["px"]), dtype=phyx_amd.rigid_body_dtype)
    for r in range(k - 1, -1, -1):
        union[slabs[r].global_index] = slabs[r].world.bodies
    fb = full.bodies
    dpos = np.hypot(union["pos"]["x"] - fb["pos"]["x"], union["pos"]["y"] - fb["pos"]["y"])
    assert dpos.max() < 0.5, dpos.max()
    nm_union, nm_full = sum(s.world.counts()[1] for s in slabs), full.counts()[1]
    assert abs(nm_union - nm_full) <= 0.03 * nm_full, (nm_union, nm_full)


def test_exchange_detects_a_diverged_or_failed_peer(built_lib):
    """The all-gather carries a real status: every segment's header holds the step serial, a status word and the topology
    fingerprint of the schedule the rank solved; the unpack flags a peer that reported a failure, is at another step,
    solved another topology, or whose segment never arrived."""
    scene = scenes.stack(6, 12)
    cfg = Configuration(phyx_amd.SOLVE_SCALAR, phyx_amd.ISLAND_MULTIPLE, 8, 8)
    from phyx_amd.dist import Exchange
    cap = Exchange.capacity_for(80, 400)
    ranks = _LocalRanks(scene, 2, cap)
    ranks.step(1.0 / 60.0, cfg)
    assert [w.solver.exchange_status() for w in ranks.worlds] == [0, 0]
    # rank 1's segment never arrives at rank 0: its slot keeps the previous step's bytes — an old serial, or no header at
    # all where the segment length changed with the schedule
    segs = [w.StepBegin(1.0 / 60.0, cfg) for w in ranks.worlds]
    for w in ranks.worlds:
        w.sync()
    ranks.recv[0].copy_from(ranks.send[0], segs[0], dst_offset=0, stream=ranks.worlds[0].stream_ptr())
    ranks.recv[1].copy_from(ranks.send[0], segs[0], dst_offset=0, stream=ranks.worlds[1].stream_ptr())
    ranks.recv[1].copy_from(ranks.send[1], segs[0], dst_offset=segs[0], stream=ranks.worlds[1].stream_ptr())
    for w in ranks.worlds:
        w.StepEnd(1.0 / 60.0)
    assert ranks.worlds[0].solver.exchange_status() & (phyx_amd.api.XCH_SERIAL_MISMATCH | phyx_amd.api.XCH_BAD_SEGMENT)
    assert ranks.worlds[1].solver.exchange_status() == 0
    # a peer posts a failure status; a zeroed slot is a segment that was never written
    ranks.worlds[1].sync()
    hdr = ranks.send[1].to_host(32).view(np.uint32).copy()
    hdr[2] = 7
    ranks.recv[1].from_host(hdr, offset=0)
    ranks.recv[1].from_host(np.zeros(8, dtype=np.uint32), offset=segs[0])
    ranks.worlds[1].StepEnd(1.0 / 60.0)
    st = ranks.worlds[1].solver.exchange_status()
    assert st & phyx_amd.api.XCH_PEER_ERROR and st & phyx_amd.api.XCH_BAD_SEGMENT
    # the unsplit entry points refuse to step a sharded world
    with pytest.raises(phyx_amd.PhxError):
        ranks.worlds[0].Update(1.0 / 60.0, cfg)


def test_two_processes_share_the_gpu_and_exchange_over_gloo(built_lib):
    """The real multi-process path (phyx_amd.dist.step_sharded + Exchange) with world_size 2: both ranks on GPU 0, the
    all-gather staged through gloo.  Each rank checks its replica against an unsharded World after every step."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    worker = os.path.join(root, "tests", "sharded_worker.py")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=420)
        assert p.returncode == 0 and "sharded worker ok" in out, out[-1500:] + err[-3000:]


def test_exchange_through_rccl_one_rank(built_lib):
    """The transport a GPU node uses — torch uint8 tensors as exchange buffers, dist.all_gather_into_tensor (backend "nccl" =
    RCCL) enqueued on the solver's own stream through an ExternalStream — with the one rank a one-GPU box can host: the sharded
    step (StepBegin / all-gather / StepEnd) against an unsharded World, and bench()'s per-step hook."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PHX_TEST_BACKEND="nccl")
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "sharded_worker.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0 and "sharded worker ok" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


def test_exchange_through_the_native_rccl_transport_one_rank(built_lib):
    """Backend "rccl": the library's own communicator (csrc/comm.hip: ncclCommInitRank / ncclAllGather resolved at run time),
    torch.distributed only hands rank 0's id around.  The Exchange path, bench() without a step hook, and the one-call native
    step phx_world_step_sharded with world-owned buffers, each against an unsharded World."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PHX_TEST_BACKEND="rccl")
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "sharded_worker.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0 and "sharded worker ok" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


def test_native_communicator_gives_up_on_a_missing_peer(built_lib):
    """csrc/comm.hip: ncclCommInitRank blocks until every rank has called it.  A rank whose peer never arrives must come back with an
    error after PHX_COMM_TIMEOUT_S seconds instead of hanging its launcher (the first multi-GPU run of the native transport is
    the driver's: it must not be able to hang)."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import phyx_amd\nfrom phyx_amd import api\nfrom phyx_amd._lib import PhxError\n"
            "assert api.Comm.rccl_version() > 0\n"
            "uid = api.Comm.unique_id()\n"
            "try:\n    api.Comm(uid, 0, 2, 0)\nexcept PhxError as e:\n    print('GAVE UP:', e); sys.stdout.flush()\n    import os; os._exit(0)\n"
            "print('created?!')\n" % root)
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PHX_COMM_TIMEOUT_S="4"), capture_output=True, text=True, timeout=300)
    assert "GAVE UP" in p.stdout and "still waits for its peers" in p.stdout, p.stdout[-1000:] + p.stderr[-2000:]
    assert time.time() - t0 < 120


@pytest.mark.parametrize("ranks", [2, 8])
@pytest.mark.parametrize("mode", ["slab", "replica"])
def test_bench_launches_its_own_ranks(built_lib, mode, ranks):
    """`python bench.py --gpus N` without torchrun: bench.py spawns the N ranks itself (here all of them on GPU 0 over gloo) and rank 0
    prints the one JSON line of the cfg-3 workload — in slab mode (ownership sharding, the default) and in replica mode; N = 8 is
    the shape of the driver's scaling run (eight processes, eight slabs / eight shares of the groups, the per-step collective)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", "--mode", mode, "--columns", "48", "--rows", "30", "--steps", "3",
                        "--warmup", "1", "--repeats", "1", "--no-secondary", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == ranks and out["steps"] == 3 and out["value"] > 0 and out["config"]["mode"] == mode and "error" not in out
    assert out["config"]["bodies_total"] == 48 * 30 + 1
    assert out["config"]["transport_info"]["ranks_seen"] == ranks
    if mode == "replica":
        assert out["extra"]["exchange"]["status"] == 0


def test_slab_worlds_are_exact_sub_worlds_and_the_guard_trips(oracle, built_lib):
    """Ownership sharding (phyx_amd.dist.SlabWorld: BASELINE config 3 read literally — every rank simulates the islands of its own
    x-slab, the per-step collective is a 4-byte all-reduce).  Three ranks emulated in one process: every slab world is the oracle's
    world of that slab, byte for byte; while no body leaves its slab the guard holds and the union agrees with the unsharded world
    (contact counts up to duplicate manifolds; positions within the spread two legal Gauss-Seidel orders have — a slab world numbers its contact points
    locally, so its colouring priorities differ); a scene whose bodies interleave across the cut trips the guard at once."""
    import types
    from phyx_amd import dist as pdist
    scene = scenes.stack(12, 20)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE, 12, 12)
    parts = pdist.slab_partition(scene, 3)
    for sub, idx, bounds in parts:                                   # each slab is a world of its own: the usual lockstep bar
        _lockstep(oracle, sub, 8, cfg, check_every=2)
    full = phyx_amd.World(0, gravity=-200.0)
    full.add_scene(scene)
    slabs = []
    for r in range(3):
        g = types.SimpleNamespace(rank=r, world_size=3, step_barrier_value=lambda v: v)
        slabs.append(pdist.SlabWorld(g, scene, device=0, gravity=-200.0))
    for _ in range(5):
        full.Update(1.0 / 60.0, cfg)
        for sw in slabs:
            sw.step(1.0 / 60.0, cfg)
    assert all(sw.inside() for sw in slabs)
    fb = full.bodies
    # (a column's boxes tie on their min x up to rounding jitter, and which of two such boxes sorts first decides whether the
    #  reference's oriented pair key creates a duplicate manifold (b, a) beside (a, b): counts agree only up to those duplicates)
    assert abs(sum(sw.world.counts()[1] for sw in slabs) - full.counts()[1]) <= 0.03 * full.counts()[1]          # manifolds
    assert abs(sum(sw.world.counts()[3] for sw in slabs) - full.counts()[3]) <= 0.03 * full.counts()[3]          # joints
    for sw in slabs:
        mine = sw.world.bodies
        dyn = mine["inv_mass"] > 0
        d = np.maximum(np.abs(mine["pos"]["x"][dyn] - fb["pos"]["x"][sw.global_index][dyn]), np.abs(mine["pos"]["y"][dyn] - fb["pos"]["y"][sw.global_index][dyn]))
        # two legal sweep orders of an unconverged 20-box stack: the reference's own scalar and AVX2 orders are 2.7e-2 apart after ONE
        # step and 0.5 after ten on such columns; five steps must stay inside that band
        assert float(d.max()) < 0.5 and float(d.mean()) < 0.1, (float(d.max()), float(d.mean()))
    one = pdist.SlabWorld(types.SimpleNamespace(rank=0, world_size=1, step_barrier_value=lambda v: v, reduce_max=float), scene, device=0, gravity=-200.0)
    one.step(1.0 / 60.0, cfg)
    assert one.gather_bodies().tobytes() == one.world.bodies.tobytes()
    # bodies of both ranks share the same stretch of the x axis: not a slab-shardable world, and the guard says so
    mixed = scenes.falling(120, width=20.0, ymax=400.0)
    g = types.SimpleNamespace(rank=0, world_size=2, step_barrier_value=lambda v: v)
    sw = pdist.SlabWorld(g, mixed, device=0, gravity=-200.0, auto_reslab=False)
    with pytest.raises(RuntimeError):
        for _ in range(30):
            sw.step(1.0 / 60.0, cfg)


def test_build_tables_are_fetched_on_demand(built_lib):
    """A speculative schedule build leaves its tables (bin offsets, classes per bin, component sizes) on the device and the step's
    settle brings back 64 bytes; the query API and the statistics fetch them when asked.  A world nobody asks must step byte for
    byte like a twin that is asked every step, and answer the same when it finally is — also right after steps whose builds
    nobody looked at."""
    scene = scenes.stack(40, 30)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE, 12, 8)
    a, b = phyx_amd.World(0, gravity=-200.0), phyx_amd.World(0, gravity=-200.0)
    a.add_scene(scene); b.add_scene(scene)
    asked = 0
    for step in range(11):
        a.Update(1.0 / 60.0, cfg); b.Update(1.0 / 60.0, cfg)
        sa, (ga, la), (oa, ca) = a.solver.stats(), a.solver.groups(), a.solver.schedule()
        if step in (0, 5, 6, 10):
            sb, (gb, lb), (ob, cb) = b.solver.stats(), b.solver.groups(), b.solver.schedule()
            for f in ("lds_islands", "colour_count", "island_count", "island_max_size", "impulse_iterations", "joint_visits", "recoloured"):
                assert getattr(sa, f) == getattr(sb, f), (step, f, getattr(sa, f), getattr(sb, f))
            assert la == lb and np.array_equal(ga, gb) and np.array_equal(oa, ob) and np.array_equal(ca, cb), step
            assert step == 0 or sb.recoloured == 2                          # (the path under test: a speculative rebuild; the first build reads its sizes back)
            asked += 1
        _same_world(b, a, "unasked world at step %d" % step)
    assert asked == 4


def test_world_lockstep_without_the_in_kernel_check(oracle, built_lib, monkeypatch):
    """PHX_NO_FUSED_VERIFY=1: no solve may check its cached schedule inside the island launch, so the world takes the paths a scene
    with more groups than fit the chip at once takes — speculative rebuilds that skip the topology-hash pass, and a rebuild with
    one when a step's joints turn out unchanged and no hash is on record.  Byte for byte against the oracle all the same."""
    monkeypatch.setenv("PHX_NO_FUSED_VERIFY", "1")
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE, 12, 8)
    _lockstep(oracle, scenes.stack(8, 30), 14, cfg)
    _lockstep(oracle, scenes.falling(300, width=200.0, ymax=260.0), 40, cfg, check_every=4)


def _emulated_slabs(scene, n, **kw):
    import types
    from phyx_amd import dist as pdist
    return [pdist.SlabWorld(types.SimpleNamespace(rank=r, world_size=n, step_barrier_value=lambda v: v), scene, device=0, gravity=-200.0, **kw) for r in range(n)]


def test_reslab_with_unchanged_cuts_hands_every_world_back_as_it_was(built_lib):
    """Re-slab (dist.SlabWorld.reslab: all-gather of the ranks' world states, new cuts, phx_world_set_state), three ranks emulated in
    one process.  On a stack the island-safe cuts fall where the first partition put them, so every rank must get back exactly the
    world it gave away — bodies, manifolds, contact points, joints, byte for byte, through global numbering, concatenation and
    re-indexing — and must then step byte for byte like a twin that never re-slabbed."""
    scene = scenes.stack(12, 20)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE, 12, 8)
    a, b = _emulated_slabs(scene, 3), _emulated_slabs(scene, 3)
    for _ in range(4):
        for sw in a + b:
            sw.world.Update(1.0 / 60.0, cfg)
    before = [sw.world.state() for sw in a]
    # the re-slab's first phase (24 bytes per dynamic body): the plan made from the ranks' intervals alone must be the cut the full
    # hand-over makes from the union world — same owners, same bounds — which is what lets a rank keep its World when nobody moves
    intervals = [sw.reslab_intervals() for sw in a]
    plans = [sw.reslab_plan(intervals) for sw in a]
    blobs = [sw.reslab_pack() for sw in a]
    for sw in a:
        sw.reslab_apply(blobs)
    for sw, (owner, gi, bounds) in zip(a, plans):
        r = sw.group.rank
        st = sw.world.bodies
        dyn = ~((st["inv_mass"] == 0) & (st["inv_inertia"] == 0))
        assert np.array_equal(np.sort(gi[owner == r]), np.sort(np.asarray(sw.global_index)[dyn]))
        assert (float(bounds[r][0]), float(bounds[r][1])) == sw.bounds
    for sw, twin, old in zip(a, b, before):
        assert sw.reslabs == 1 and np.array_equal(sw.global_index, twin.global_index)
        assert sw.bounds[0] <= twin.bounds[0] + 10 and sw.inside()
        for got, want, what in zip(sw.world.state(), old, ("bodies", "manifolds", "contact points", "joints")):
            assert got.tobytes() == want.tobytes(), what
    for step in range(5):
        for sw, twin in zip(a, b):
            sw.world.Update(1.0 / 60.0, cfg); twin.world.Update(1.0 / 60.0, cfg)
            _same_world(sw.world, twin.world, "re-slabbed rank %d at step %d" % (sw.group.rank, step))
    # one rank: the collective form (reslab(): pack, all-gather, apply) over the single-process group is the identity too
    from phyx_amd import dist as pdist
    one = pdist.SlabWorld(pdist.Single(), scene, device=0, gravity=-200.0, reslab_every=2)
    twin = phyx_amd.World(0, gravity=-200.0)
    twin.add_scene(scene)
    for step in range(5):
        one.step(1.0 / 60.0, cfg); twin.Update(1.0 / 60.0, cfg)
        _same_world(one.world, twin, "single-rank slab world at step %d" % step)
    assert one.reslabs == 2 and one.reslabs_in_place == 2 and one.inside()      # (nobody moved: the World was kept both times)


def test_reslab_follows_piles_that_grow_into_each_other(built_lib):
    """Six separate piles on three ranks (scenes.piles): they widen as they settle, reach across the cuts between the ranks' slabs and
    merge — the guard trips, the ranks re-slab, and the world goes on.  After every hand-over each guard holds, every dynamic body is
    on exactly one rank, the union of the ranks' bodies (gather order) is what the ranks held before it, and nothing blows up; a
    rank may end up without bodies (fewer separable piles than ranks) and keeps stepping."""
    scene = scenes.piles(6, 60, pitch=64.0, ymax=220.0)
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE, 10, 6)
    slabs = _emulated_slabs(scene, 3)
    n_dyn = int(np.count_nonzero(~scene["static"]))
    tripped = 0
    for step in range(160):
        for sw in slabs:
            sw.world.Update(1.0 / 60.0, cfg)
        bad = [not sw.inside() for sw in slabs]
        if any(bad) or step % 32 == 31:
            tripped += any(bad)
            union_before = np.zeros(len(scene["px"]), dtype=phyx_amd.rigid_body_dtype)
            for sw in slabs:
                union_before[sw.global_index] = sw.world.bodies
            blobs = [sw.reslab_pack() for sw in slabs]
            for sw in slabs:
                sw.reslab_apply(blobs)
            owners = np.zeros(len(scene["px"]), dtype=int)
            union_after = np.zeros_like(union_before)
            for sw in slabs:
                mine = sw.world.bodies
                owners[sw.global_index[mine["inv_mass"] > 0]] += 1
                union_after[sw.global_index] = mine
                assert sw.inside(), (step, sw.group.rank, sw.bounds)
            assert np.array_equal(owners[~scene["static"]], np.ones(n_dyn, dtype=int))
            union_before["index"] = 0; union_after["index"] = 0          # (a body's own index is local to its world)
            assert union_after.tobytes() == union_before.tobytes()
            assert sum(sw.world.counts()[3] for sw in slabs) > 0
    assert tripped >= 1, "the piles never reached a slab boundary: the scene does not exercise the guard"
    for sw in slabs:
        pos = sw.world.bodies["pos"]
        assert np.isfinite(pos["x"]).all() and np.isfinite(pos["y"]).all() and float(pos["y"].min()) > -50.0


@pytest.mark.parametrize("ranks,every", [(2, 0), (3, 5)])
def test_reslab_across_processes(built_lib, ranks, every):
    """The same hand-over between real processes (tools/reslab_ranks.py: one process per rank, all on GPU 0, collectives over gloo):
    the guard's verdict is all-reduced, the states are all-gathered, every rank restores its new slab."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "reslab_ranks.py"), "--ranks", str(ranks), "--backend", "gloo", "--steps", "160", "--every", str(every)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["ranks"] == ranks and out["steps"] == 160 and out["reslabs"] >= 1
    assert out["dynamic_bodies_total"] == out["dynamic_bodies_scene"] and out["every_guard_holds"] and out["finite"]
    # the library's re-slab (phx_world_reslab, csrc/reslab.hip; here over the group's collectives as its host callbacks) against the numpy
    # statement of the same hand-over: the whole world after 160 steps and every re-slab on the way, byte for byte
    q = subprocess.run([sys.executable, os.path.join(root, "tools", "reslab_ranks.py"), "--ranks", str(ranks), "--backend", "gloo", "--steps", "160", "--every", str(every), "--python"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert q.returncode == 0, q.stdout[-1500:] + q.stderr[-3000:]
    twin = json.loads([ln for ln in q.stdout.splitlines() if ln.startswith("{")][-1])
    assert twin["digest"] == out["digest"] and twin["reslabs"] == out["reslabs"]


def test_world_state_save_and_restore_is_exact(built_lib):
    """Checkpoint / resume (phx_world_set_state): a fresh world restored from what the four getters return — bodies, manifolds with
    their contact points, joints with their warm-start impulses; the broadphase's pair set is rebuilt from the manifolds — steps
    byte for byte like the world the state was taken from, through contacts dying and new ones appearing."""
    scene = scenes.falling(900, width=120.0, ymax=320.0)
    cfg = Configuration(phyx_amd.SOLVE_SCALAR, phyx_amd.ISLAND_MULTIPLE, 12, 8)
    a = phyx_amd.World(0, gravity=-200.0)
    a.add_scene(scene)
    for _ in range(30):
        a.Update(1.0 / 60.0, cfg)
    saved = a.state()
    assert len(saved[1]) > 100 and len(saved[3]) > 100
    b = phyx_amd.World(0, gravity=-200.0)                                    # never saw the scene
    b.set_state(*saved)
    assert b.counts() == a.counts()
    for got, want in zip(b.state(), saved):
        assert got.tobytes() == want.tobytes()
    births = deaths = 0
    for step in range(25):
        before = a.counts()[1]
        a.Update(1.0 / 60.0, cfg)
        b.Update(1.0 / 60.0, cfg)
        _same_world(b, a, "restored world at step %d" % step)
        assert b.contactPoints.tobytes() == a.contactPoints.tobytes()
        births += a.collider.stats().new_pairs
        deaths += max(0, before + a.collider.stats().new_pairs - a.counts()[1])
    assert births > 0 and deaths > 0                                         # the pair set was exercised both ways
    # a state that does not hang together is refused
    bodies, manifolds, cps, joints = (x.copy() for x in saved)
    bad = joints.copy(); bad["body1"][0] += 1
    with pytest.raises(phyx_amd.PhxError):
        b.set_state(bodies, manifolds, cps, bad)
    with pytest.raises(phyx_amd.PhxError):
        b.set_state(bodies, manifolds, cps[:-1], joints)
    b.set_state(*saved)                                                      # and the handle is still usable
    assert b.counts() == (len(bodies), len(manifolds), len(cps), len(joints))


def test_uploaded_accelerations_are_applied_once(oracle, built_lib):
    """Records handed to phx_world_set_state may carry accelerations (the reference's demo sets them on a dragged body, ref:
    main.cpp:345-346); IntegrateVelocity applies them once and zeroes them (ref: World.cpp:44-53).  A restored world must do the
    same — the resident arrays carry no accelerations, so the first step after the upload takes them from a one-shot table —
    byte for byte with the oracle world whose records were given the same accelerations."""
    scene = scenes.stack(3, 14)
    cfg = Configuration(phyx_amd.SOLVE_SCALAR, phyx_amd.ISLAND_MULTIPLE, 10, 10)
    pw, ow = _lockstep(oracle, scene, 3, cfg)
    bodies, manifolds, cps, joints = (x.copy() for x in pw.state())
    rng = np.random.default_rng(3)
    n = len(bodies)
    ax, ay, aa = (rng.normal(0, 40, n).astype(np.float32) for _ in range(3))
    ax[0] = ay[0] = aa[0] = 0                                              # (the static ground)
    bodies["acceleration"]["x"] = ax; bodies["acceleration"]["y"] = ay; bodies["angular_acceleration"] = aa
    live = ow.bodies()                                                       # the oracle world's own records
    live["acceleration"]["x"] = ax; live["acceleration"]["y"] = ay; live["angular_acceleration"] = aa
    w = phyx_amd.World(0, gravity=-200.0)
    w.set_state(bodies, manifolds, cps, joints)
    assert w.bodies.tobytes() == ow.bodies().tobytes()
    for step in range(3):
        w.Update(1.0 / 60.0, cfg)
        ow.pre_solve(1.0 / 60.0)
        order, offs = w.solver.schedule()
        groups, _ = w.solver.groups()
        oracle.solver_solve_grouped(ow.bodies(), ow.contact_points(), ow.joints(), order, offs, groups, 10, 10, oracle.STAG_COLOUR_SYNC)
        ow.integrate_position(1.0 / 60.0)
        assert w.bodies.tobytes() == ow.bodies().tobytes(), "bodies differ at step %d after the upload" % step
        assert not w.bodies["acceleration"]["x"].any() and not w.bodies["angular_acceleration"].any()
        assert w.contactJoints.tobytes() == ow.joints().tobytes()


def test_update_is_queued_and_getters_synchronise(oracle, built_lib):
    """phx_world_update returns once the step is queued on the world's stream; every getter waits for it.  Two worlds, one
    read after every step, one only at the end (with an explicit synchronize), must agree bit for bit; the per-phase timers
    are off unless asked for."""
    cfg = Configuration(phyx_amd.SOLVE_SCALAR, phyx_amd.ISLAND_MULTIPLE, 10, 10)
    eager, lazy = phyx_amd.World(0, gravity=-200.0), phyx_amd.World(0, gravity=-200.0)
    for w in (eager, lazy):
        w.add_scene(scenes.stack(12, 40))
    for _ in range(12):
        eager.Update(1.0 / 60.0, cfg)
        _ = eager.bodies, eager.contactJoints, eager.solver.stats()
        lazy.Update(1.0 / 60.0, cfg)
    lazy.sync()
    assert eager.bodies.tobytes() == lazy.bodies.tobytes() and eager.contactJoints.tobytes() == lazy.contactJoints.tobytes()
    assert eager.counts() == lazy.counts() and eager.counts()[3] > 0
    assert sum(lazy.phase_ms().values()) == 0.0                       # timers never switched on
    lazy.set_phase_timing(True)
    lazy.Update(1.0 / 60.0, cfg)
    ph = lazy.phase_ms()
    assert ph["SolveJoints"] > 0.0 and ph["UpdatePairs"] > 0.0
    eager.Update(1.0 / 60.0, cfg)
    assert eager.bodies.tobytes() == lazy.bodies.tobytes()            # timing changes nothing but the waits


def test_world_api_errors(built_lib):
    w = phyx_amd.World(0)
    with pytest.raises(phyx_amd.PhxError):
        w.AddBody((0, 0), 0.0, (0.0, 1.0))
    with pytest.raises(phyx_amd.PhxError):
        w.set_shard(3, 2)
    w.Update(1.0 / 60.0, Configuration())          # empty world steps fine
    assert w.counts() == (0, 0, 0, 0)


def test_debugging_knobs_do_not_change_results(built_lib):
    """The readback mailbox, the speculative solve / deferred build check and the device schedule builder are performance
    mechanisms — like the fused launches of partitioned components, the second stream and the in-kernel schedule check: with each of
    them switched off (PHX_NO_MAILBOX, PHX_NO_SPECULATION, PHX_SCHEDULE_BUILDER=host, PHX_NO_SPEC_BINS, PHX_NO_PARTS, PHX_NO_SIDE_STREAM,
    PHX_NO_FUSED_VERIFY, PHX_NO_SPLIT_SORT, PHX_NO_MAIL_CARRIER — the mailbox posts riding in the next kernel's first workgroup —, PHX_NO_TAIL —
    the HBM group's trailing tiny classes one launch each instead of one workgroup's launch —, PHX_NO_JP_WALK_ONE — its colouring walk one launch per
    round with the host's look in between instead of one workgroup's launch) a world that
    rebuilds its schedule every step, merges islands and falls back to the host builder (a 90-box clique) must produce the very
    same bytes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def digest(extra_env, mode):
        env = dict(os.environ)
        env.update(extra_env)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "knob_check.py"), str(mode)], cwd=root, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        return r.stdout.strip().splitlines()[-1]
    for mode in (phyx_amd.ISLAND_SINGLE, phyx_amd.ISLAND_MULTIPLE_SLOPPY):
        want = digest({}, mode)
        assert len(want) == 64
        for knob in ({"PHX_NO_MAILBOX": "1"}, {"PHX_NO_SPECULATION": "1"}, {"PHX_SCHEDULE_BUILDER": "host"}, {"PHX_NO_SPEC_BINS": "1"},
                     {"PHX_NO_PARTS": "1"}, {"PHX_NO_SIDE_STREAM": "1"}, {"PHX_NO_FUSED_VERIFY": "1"}, {"PHX_NO_SPLIT_SORT": "1"}, {"PHX_NO_MAIL_CARRIER": "1"},
                     {"PHX_NO_PRELABEL": "1"}, {"PHX_NO_TAIL": "1"}, {"PHX_NO_JP_WALK_ONE": "1"}):
            assert digest(knob, mode) == want, (knob, mode)


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["stack", "merge", "falling", "tilted"])
def test_rebuild_from_the_manifolds_builds_the_same_schedule(built_lib, scene):
    """The World's schedule rebuild takes its connected components, their joint counts and the bins from the MANIFOLDS, on the side
    stream, while the joint list is still being refreshed (csrc/schedule_kernels.h k_cc_link_manifolds, k_manifold_components), pairs the
    joints into units through ContactPoint::solverIndex and deals them to their bins by a fill counter (k_joint_scatter<true>); it must
    build the schedule the rebuild from the joints builds — the schedule is a pure function of the joints.  tools/build_twin.py hashes
    the schedule (slot order, class offsets, groups) and every array after every step; with PHX_NO_PRELABEL=1 the digest is the same,
    and without it the stacks' rebuilds really do come from the manifolds most of the time."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(extra):
        env = dict(os.environ)
        env.update(extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "build_twin.py"), scene, "45"], cwd=root, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        digest, lite, full = r.stdout.strip().splitlines()[-1].split()
        return digest, int(lite), int(full)
    d_inc, lite, full = run({})
    d_full, lite0, full0 = run({"PHX_NO_PRELABEL": "1"})
    assert d_inc == d_full
    assert lite0 == 0 and full0 >= lite + full - 2
    if scene in ("stack", "merge"):
        assert lite >= 10, (lite, full)
