"""Parity of the HIP broadphase (key build, stable radix sort, entry gather, sweep, persistent pair set)
against the oracle on MI355X.  Integer / index work: everything must match exactly."""
import numpy as np
import pytest

import phyx_amd
from phyx_amd import scenes
from helpers import oracle_world

pytestmark = pytest.mark.gpu


@pytest.fixture()
def collider(built_lib):
    return phyx_amd.Collider(0)


def _bodies(scene, steps=0, oracle=None):
    w = oracle_world(scene)
    for _ in range(steps):
        w.update()
    return w.bodies().copy()


def _check_update(collider, oracle, bodies, known_pairs):
    """one UpdateBroadphase+UpdatePairs: sorted order, entries, and the new-pair list in emission order"""
    keys, srt, ent = oracle.broadphase_build(bodies)
    cand, cnt, tests = oracle.sweep_candidates(ent)
    want_new = [tuple(p) for p in cand.tolist() if tuple(p) not in known_pairs]
    got_new = collider.UpdateBroadphaseAndPairs(bodies)
    gs, ge = collider.sorted(len(bodies))
    assert gs.tobytes() == srt.tobytes(), "sorted {key,index} sequence differs (radix sort / radixFloat)"
    assert ge.tobytes() == ent.tobytes(), "BroadphaseEntry records differ"
    assert [tuple(p) for p in got_new.tolist()] == want_new, "new pairs or their emission order differ"
    st = collider.stats()
    assert st.candidate_tests == tests and st.overlapping_pairs == cnt and st.new_pairs == len(want_new)
    known_pairs.update(want_new)
    assert st.set_size == len(known_pairs)
    return got_new


@pytest.mark.parametrize("scene", ["stack2x10", "stack10x100", "falling", "tilted", "wide"])
def test_first_update_matches_oracle(collider, oracle, scene):
    sc = {"stack2x10": lambda: scenes.stack(2, 10), "stack10x100": lambda: scenes.stack(10, 100),
          "falling": lambda: scenes.falling(3000, width=200.0, ymax=300.0), "tilted": lambda: scenes.tilted(200),
          "wide": lambda: scenes.stack(700, 3)}[scene]()
    _check_update(collider, oracle, _bodies(sc), set())


def test_persistent_set_over_steps(collider, oracle):
    """Pairs found in earlier steps are not reported again; erased pairs come back (ref: Collider.cpp:313, :391)."""
    w = oracle_world(scenes.falling(800, width=80.0, ymax=300.0))
    known = set()
    for step in range(40):
        w.pre_solve()
        new = _check_update(collider, oracle, w.bodies().copy(), known)
        assert [tuple(p) for p in new.tolist()] == [tuple(p) for p in w.new_pairs().tolist()]
        # mirror PackManifolds' erasures: whatever the oracle world no longer holds leaves the set
        alive = set(zip(w.manifolds()["body1"].tolist(), w.manifolds()["body2"].tolist()))
        gone = [p for p in known if p not in alive]
        if gone:
            collider.erase(np.array(gone, dtype=np.uint32))
            known.difference_update(gone)
        w.solve_and_integrate()
    assert len(known) > 500
    collider.clear()
    assert len(collider.UpdateBroadphaseAndPairs(w.bodies().copy())) >= len(known) * 0.5


def test_ties_negative_keys_and_edge_sizes(collider, oracle):
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 5000):
        b = np.zeros(n, dtype=phyx_amd.rigid_body_dtype)
        minx = rng.choice(np.array([-7507.5, -15.0, -0.0, 0.0, 5.0, 20.0, 1e-30, -1e-30, 3e38], dtype=np.float32), size=n)
        b["aabb_min"]["x"] = minx
        b["aabb_max"]["x"] = minx + rng.uniform(0, 3, n).astype(np.float32)
        b["aabb_min"]["y"] = rng.uniform(-50, 50, n).astype(np.float32)
        b["aabb_max"]["y"] = b["aabb_min"]["y"] + rng.uniform(0, 30, n).astype(np.float32)
        collider.clear()
        _check_update(collider, oracle, b, set())


def test_hub_rows(collider, oracle):
    """A body whose AABB spans thousands of others (the ground) takes the workgroup-per-row path."""
    sc = scenes.stack(1200, 8)                      # ground + 9600 boxes: ground row scans 9600 candidates
    b = _bodies(sc)
    _check_update(collider, oracle, b, set())
    assert collider.stats().candidate_tests > 9600


def test_full_size_1m_boxes_properties(collider, oracle):
    """BASELINE config 4 size (1 000 001 bodies, stack(10000,100)): exact against the oracle's sort, and the
    sweep through its size-independent properties (every emitted pair overlaps, counts match the oracle)."""
    sc = scenes.stack(10000, 100)
    b = _bodies(sc)
    keys, srt, ent = oracle.broadphase_build(b)
    new = collider.UpdateBroadphaseAndPairs(b)
    gs, ge = collider.sorted(len(b))
    assert gs.tobytes() == srt.tobytes() and ge.tobytes() == ent.tobytes()
    assert (np.diff(gs["value"].astype(np.int64)) >= 0).all()
    cand, cnt, tests = oracle.sweep_candidates(ent)                        # the reference's serial sweep (ref: Collider.cpp:296-318): every pair, in its order
    st = collider.stats()
    assert st.overlapping_pairs == cnt == len(new) and st.candidate_tests == tests
    assert new.tobytes() == np.ascontiguousarray(cand, dtype=np.uint32).tobytes(), "the 1M-body pair list differs from the serial sweep's, element by element"

    a, c = b[new[:, 0]], b[new[:, 1]]
    assert (a["aabb_min"]["x"] <= c["aabb_max"]["x"]).all() and (c["aabb_min"]["x"] <= a["aabb_max"]["x"]).all()
    # idempotence: a second update over the same bodies reports nothing new — and it takes the two-level sort (splitters of the
    # first update on record: csrc/splitter_sort.h), which must leave the very same sequence
    assert len(collider.UpdateBroadphaseAndPairs(b)) == 0
    gs2, ge2 = collider.sorted(len(b))
    assert gs2.tobytes() == srt.tobytes() and ge2.tobytes() == ent.tobytes()


def _random_boxes(rng, n, minx):
    b = np.zeros(n, dtype=phyx_amd.rigid_body_dtype)
    b["aabb_min"]["x"] = minx
    b["aabb_max"]["x"] = minx + rng.uniform(0, 3, n).astype(np.float32)
    b["aabb_min"]["y"] = rng.uniform(-50, 50, n).astype(np.float32)
    b["aabb_max"]["y"] = b["aabb_min"]["y"] + rng.uniform(0, 30, n).astype(np.float32)
    return b


def test_two_level_sort_with_fresh_stale_and_useless_splitters(collider, oracle):
    """csrc/splitter_sort.h: from its second update on (same body count) a Collider deals the bodies into buckets by the previous
    update's splitters and sorts the buckets in LDS.  The splitters only balance the work: the sorted sequence must equal the
    reference's stable radix sort whatever they are — bodies that moved a little (the normal case), a completely different scene of
    the same size (buckets of thousands of records: the workgroup's in-HBM network), all keys equal (ties resolved by the index)."""
    rng = np.random.default_rng(17)
    n = 20000
    x = np.sort(rng.uniform(-5000, 5000, n)).astype(np.float32)
    _check_update(collider, oracle, _random_boxes(rng, n, x), set())                 # LSD sort, leaves splitters
    collider.clear()
    _check_update(collider, oracle, _random_boxes(rng, n, x + rng.normal(0, 2.0, n).astype(np.float32)), set())    # moved a little
    collider.clear()
    _check_update(collider, oracle, _random_boxes(rng, n, x[::-1].copy()), set())     # index order reversed against the splitters
    collider.clear()
    _check_update(collider, oracle, _random_boxes(rng, n, np.full(n, 7.25, np.float32) + (rng.integers(0, 2, n) * 1e-3).astype(np.float32)), set())   # two key values: most bodies in two buckets
    collider.clear()
    _check_update(collider, oracle, _random_boxes(rng, n, np.full(n, -3.5, np.float32)), set())   # after an unbalanced update: the LSD sort again
    collider.clear()
    _check_update(collider, oracle, _random_boxes(rng, n, np.full(n, -3.5, np.float32)), set())   # every key equal, splitters by index
    collider.clear()
    _check_update(collider, oracle, _random_boxes(rng, n, rng.uniform(-1e4, 1e4, n).astype(np.float32)), set())


@pytest.mark.parametrize("n", [255, 256, 257, 512, 513, 767, 769, 4095, 4097, 12289])
def test_two_level_sort_edge_sizes(collider, oracle, n):
    """bucket counts 1, 2, 3, ... — records at the bucket boundaries, the last bucket short"""
    rng = np.random.default_rng(n)
    vals = np.array([-7507.5, -15.0, -0.0, 0.0, 5.0, 20.0, 1e-30, -1e-30, 3e38], dtype=np.float32)
    for rep in range(3):
        collider.clear()
        minx = np.where(rng.random(n) < 0.3, rng.choice(vals, size=n), rng.uniform(-100, 100, n)).astype(np.float32)
        _check_update(collider, oracle, _random_boxes(rng, n, minx), set())
