"""The N>1 plumbing of bench.py on CPU: two processes over gloo (the GPU box only ever runs world_size 1 here;
the 8-GPU run belongs to the driver).  Covers rendezvous on 127.0.0.1, the per-step barrier, max/sum reductions
and the slab partition each rank derives for itself."""
import os
import numpy as np
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from phyx_amd import dist as pdist
    g = pdist.init(2, backend="gloo")
    first, n = pdist.shard_columns(2001, g.rank, g.world_size)
    g.barrier()
    flags = [g.step_barrier() for _ in range(3)]           # the per-step 4-byte all-reduce
    hook = g.stream_hook(0)                                 # bench.py's per-step hook (gloo: the blocking all-reduce)
    for step in range(3):
        hook(step, 1); hook(step, 0)
    import numpy as np
    seg = np.full(512, 10 + g.rank, dtype=np.uint8)         # the host-staged all-gather of the island-sharded exchange
    seg[:4] = np.frombuffer(np.uint32(0x45584850).tobytes(), dtype=np.uint8)
    allb = g.all_gather_bytes(seg)
    assert allb.shape == (1024,) and (allb[4:512] == 10).all() and (allb[516:] == 11).all()
    assert allb[:4].tobytes() == allb[512:516].tobytes() == b"PHXE"
    blobs = pdist.all_gather_blobs(g, pdist._blob(np.arange(3 + 5 * g.rank, dtype=np.int64), np.full(2, g.rank, dtype=np.float32)))   # the re-slab's all-gather: byte strings of different lengths
    for r, b in enumerate(blobs):
        idx, tag = pdist._unblob(b, (np.int64, np.float32))
        assert len(idx) == 3 + 5 * r and idx[-1] == 2 + 5 * r and (tag == r).all()
    verdicts = [g.step_barrier_value(1 if (g.rank == 1 and k == 1) else 0) for k in range(3)]   # the slab guard's verdict reaches every rank
    assert verdicts == [0, 1, 0]
    t = g.reduce_max(1.0 + g.rank)                          # max over ranks (timing)
    units = g.reduce_sum(float(n))                          # whole-job units
    print(json.dumps({"rank": g.rank, "world": g.world_size, "first": first, "n": n, "t": t, "units": units, "flags": flags}))
    g.shutdown()
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(180)
def test_two_ranks_over_gloo():
    import json
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=150)
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    assert [o["world"] for o in outs] == [2, 2]
    assert (outs[0]["first"], outs[0]["n"]) == (0, 1001) and (outs[1]["first"], outs[1]["n"]) == (1001, 1000)
    assert all(o["t"] == 2.0 and o["units"] == 2001.0 and o["flags"] == [0, 0, 0] for o in outs)


def test_shard_columns_partition():
    from phyx_amd.dist import shard_columns
    for total in (1, 7, 8, 1000, 8003):
        for ws in (1, 2, 3, 8):
            parts = [shard_columns(total, r, ws) for r in range(ws)]
            assert sum(n for _, n in parts) == total
            nxt = 0
            for first, n in parts:
                assert first == nxt and n >= total // ws
                nxt = first + n


def test_single_process_group_is_torch_free():
    code = "import sys; sys.path.insert(0, %r); from phyx_amd import dist; g = dist.init(1); g.barrier(); g.step_barrier(); assert g.stream_hook(0) is None; assert g.reduce_sum(3) == 3.0; assert 'torch' not in sys.modules" % ROOT
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    subprocess.check_call([sys.executable, "-c", code], env=env)


@pytest.mark.timeout(120)
def test_self_launch_spawns_one_process_per_rank(tmp_path):
    """`bench.py --gpus N` started without a launcher spawns its own ranks (phyx_amd.dist.self_launch): every child gets RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* and the same command line; the exit status is the worst child's."""
    script = tmp_path / "child.py"
    script.write_text("import os, sys\n"
                      "open(os.path.join(sys.argv[1], 'rank%s' % os.environ['RANK']), 'w').write(' '.join(os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))\n"
                      "sys.exit(3 if os.environ['RANK'] == '1' and len(sys.argv) > 2 else 0)\n")
    from phyx_amd.dist import self_launch
    env = dict(os.environ)
    os.environ.pop("WORLD_SIZE", None)
    try:
        assert self_launch(3, [str(script), str(tmp_path)]) == 0
        got = sorted(p.name for p in tmp_path.glob("rank*"))
        assert got == ["rank0", "rank1", "rank2"]
        fields = [(tmp_path / n).read_text().split() for n in got]
        assert [f[0] for f in fields] == ["0", "1", "2"] and all(f[2] == "3" and f[3] == "127.0.0.1" for f in fields)
        assert len({f[4] for f in fields}) == 1
        assert self_launch(2, [str(script), str(tmp_path), "fail"]) == 3
    finally:
        os.environ.clear(); os.environ.update(env)


def test_slab_partition_cuts_between_columns():
    """Ownership sharding (phyx_amd.dist.slab_partition): equal shares of the dynamic bodies by x, never a cut through bodies with the
    same centre (a column of a stack), static bodies in every slab, original relative order kept, slab intervals that tile the axis."""
    import numpy as np
    from phyx_amd import scenes
    from phyx_amd.dist import slab_partition
    for scene, n in ((scenes.stack(10, 5), 4), (scenes.stack(1000, 3), 8), (scenes.falling(300), 3), (scenes.stack(3, 4), 1), (scenes.stack(2, 6), 4)):
        parts = slab_partition(scene, n)
        assert len(parts) == n
        total = len(scene["px"])
        static = np.flatnonzero(scene["static"])
        seen = np.zeros(total, dtype=int)
        prev_hi = -np.inf
        for sub, idx, (lo, hi) in parts:
            assert np.all(np.diff(idx) > 0)                                   # original order
            assert set(static.tolist()) <= set(idx.tolist())                  # every static body in every slab
            assert np.array_equal(sub["px"], scene["px"][idx]) and np.array_equal(sub["static"], scene["static"][idx])
            dyn = idx[~scene["static"][idx]]
            seen[dyn] += 1
            if len(dyn):
                assert lo == prev_hi or (lo == -np.inf and prev_hi == -np.inf)
                assert np.all(scene["px"][dyn] > lo) and np.all(scene["px"][dyn] < hi)
                prev_hi = hi
        assert np.all(seen[~scene["static"]] == 1)                            # every dynamic body in exactly one slab
    sizes = [len(p[1]) - 1 for p in slab_partition(scenes.stack(1000, 3), 8)]
    assert max(sizes) - min(sizes) <= 3                                       # whole columns: at most one column of imbalance


def test_slab_cuts_keep_touching_bodies_together():
    """dist.slab_cuts (the re-slab's partition): bodies whose x-intervals overlap are never separated, the cuts sit inside the gaps,
    every body has exactly one owner, counts are as even as the gaps allow, and fewer blocks than ranks leaves ranks empty."""
    from phyx_amd import dist as pdist
    rng = np.random.default_rng(5)
    for trial in range(200):
        n = int(rng.integers(1, 60)); ranks = int(rng.integers(1, 6)); margin = float(rng.choice([0.0, 0.5, 2.0]))
        lo = np.sort(rng.uniform(0, 300, n)) if trial % 2 else rng.uniform(0, 300, n)
        hi = lo + rng.uniform(1, 12, n)
        owner, bounds = pdist.slab_cuts(lo, hi, ranks, margin)
        assert len(bounds) == ranks and owner.min() >= 0 and owner.max() < ranks
        for i in range(n):
            a, b = bounds[owner[i]]
            assert a < lo[i] - margin / 2 or a == -np.inf
            assert b > hi[i] + margin / 2 or b == np.inf
            touching = (lo - margin <= hi[i] + margin) & (hi + margin >= lo[i] - margin)
            assert np.all(owner[touching] == owner[i])
        live = [b for b in bounds if b[0] != np.inf]
        assert live[0][0] == -np.inf and live[-1][1] == np.inf
        for (a0, b0), (a1, b1) in zip(live, live[1:]):
            assert b0 == a1                                        # the slabs tile the axis
    owner, bounds = pdist.slab_cuts([0.0, 100.0], [5.0, 105.0], 4)
    assert sorted(set(owner)) == [0, 1] or len(set(owner)) == 2
    assert sum(1 for b in bounds if b[0] == np.inf) == 2


def test_state_blobs_round_trip():
    from phyx_amd import dist as pdist
    import phyx_amd
    arrays = (np.arange(5, dtype=np.int64), np.zeros(3, dtype=phyx_amd.rigid_body_dtype), np.zeros(0, dtype=phyx_amd.manifold_dtype),
              np.ones(7, dtype=phyx_amd.contact_joint_dtype))
    arrays[1]["pos"]["x"] = [1.5, 2.5, -3.0]
    back = pdist._unblob(pdist._blob(*arrays), [a.dtype for a in arrays])
    for a, b in zip(arrays, back):
        assert a.dtype == b.dtype and a.tobytes() == b.tobytes()
    assert [len(x) for x in pdist.all_gather_blobs(pdist.Single(), pdist._blob(*arrays))] == [len(pdist._blob(*arrays))]


def test_library_slab_cuts_and_plan_equal_the_numpy_statement(built_lib):
    """csrc/reslab.hip's planning (host code: no device needed) against phyx_amd.dist.slab_cuts / SlabWorld.reslab_plan on random
    intervals: ties, nested and touching intervals, fewer blocks than ranks, no body at all."""
    import ctypes as C
    from phyx_amd import dist
    rng = np.random.default_rng(3)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for trial in range(400):
        n = int(rng.integers(0, 60)); nr = int(rng.integers(1, 6)); margin = float(rng.choice([0.0, 0.5, 1.0]))
        lo = rng.uniform(-100, 100, n)
        if n and rng.random() < 0.4:
            lo = np.round(lo / 10) * 10                                  # ties
        hi = lo + rng.uniform(0.1, 15 if rng.random() < 0.8 else 120, n)
        owner_py, bounds_py = dist.slab_cuts(lo, hi, nr, margin)
        owner = np.zeros(max(n, 1), dtype=np.int32); bounds = np.zeros(2 * nr)
        assert built_lib.phx_reslab_cuts(vp(lo), vp(hi), n, nr, margin, vp(owner), vp(bounds)) == 0
        assert np.array_equal(owner_py, owner[:n]) and np.array_equal(np.array(bounds_py, dtype=np.float64).reshape(-1), bounds), trial
        # the plan: every rank's intervals in any order -> sorted by scene index, owners aligned with that order
        gi = rng.permutation(1000)[:n].astype(np.int64)
        order = np.argsort(gi, kind="stable")
        g2, l2, h2 = gi.copy(), lo.copy(), hi.copy()
        assert built_lib.phx_reslab_plan(vp(g2), vp(l2), vp(h2), n, nr, margin, vp(owner), vp(bounds)) == 0
        want_owner, want_bounds = dist.slab_cuts(lo[order], hi[order], nr, margin)
        assert np.array_equal(g2, gi[order]) and np.array_equal(l2, lo[order]) and np.array_equal(owner[:n], want_owner)
        assert np.array_equal(np.array(want_bounds, dtype=np.float64).reshape(-1), bounds)
