"""Host-side logic of the product that needs no GPU: scene generators, the colour schedule and the island
partition (C++ code inside libphyx_amd.so, reached through its host-only C-ABI entry points)."""
import numpy as np
import pytest

import phyx_amd
from phyx_amd import scenes
from helpers import SMALL_SCENES, presolve_state, is_static


def test_stack_scene_shape():
    s = scenes.stack(3, 4)
    assert len(s["px"]) == 13 and s["static"][0] and not s["static"][1:].any()
    assert s["sx"][0] == 45.0 and s["sy"][0] == 10.0
    assert list(s["py"][1:5]) == [15.0, 25.0, 35.0, 45.0]
    assert sorted(set(s["px"][1:].tolist())) == [-15.0, 0.0, 15.0]
    shifted = scenes.stack(3, 4, x_offset_columns=10)
    assert np.allclose(shifted["px"][1:] - s["px"][1:], 150.0)


def test_falling_scene_is_reproducible():
    a, b = scenes.falling(100, seed=3), scenes.falling(100, seed=3)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c = scenes.falling(100, seed=4)
    assert not np.array_equal(a["px"], c["px"])
    assert (np.abs(a["px"][1:]) <= 500).all() and (a["py"][1:] >= 50).all() and (a["py"][1:] <= 1000).all()


def _priority(pid, j, lower):
    """csrc/schedule.h colour_priority restated: units whose LOWER body index is even rank above the odd ones (bit 63), inside a parity
    31 bits of a multiplicative hash of the priority id, the joint index breaks ties; never zero."""
    x = (pid * 0x9E3779B1) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    return ((((~lower) & 1) << 63) | ((x >> 1) << 32) | j) + 1


@pytest.mark.parametrize("ids", ["joint_index", "contact_point_index"])
@pytest.mark.parametrize("name", list(SMALL_SCENES) + ["synthetic_units_and_near_misses", "wall48x60", "synthetic_one_big_component"])
def test_colour_schedule_invariants(built_lib, name, ids):
    """The schedule rule of csrc/schedule.h restated independently in Python: units (the two joints of a body pair whose ids
    differ in the lowest bit), first fit over the units in priority order with two candidates, the choice per connected
    component, the interior / boundary kinds of a component of more than 1024 joints (the wall: one component of 1.1e4), and
    the layout of a class: leaders that have a follower, single leaders, followers in their leaders' order."""
    if name in SMALL_SCENES:
        make, warm = SMALL_SCENES[name]
        bodies, _, joints = presolve_state(make(), warm)
    elif name == "wall48x60":
        bodies, _, joints = presolve_state(scenes.wall(48, 60), 10)      # (the rows come to rest on each other from the bottom up: one component by step 10)
    elif name == "synthetic_one_big_component":                      # random pairs over 1500 bodies: one component of thousands of joints,
        from test_solver_gpu import _random_state                    # its units of all three kinds (interior at level 0 / 1, rest)
        bodies, _, joints = _random_state(np.random.default_rng(11), 1500, 5000, 0.02, units=True)
    else:                                                            # couples on random body pairs + the near misses that must not pair
        from test_solver_gpu import _random_state
        bodies, _, joints = _random_state(np.random.default_rng(6), 300, 900, 0.1, units=True)
    static = is_static(bodies)
    nj = len(joints)
    pid = np.arange(nj, dtype=np.int32) if ids == "joint_index" else joints["contact_point_index"].astype(np.int32)
    order, offs = phyx_amd.schedule_colours(joints["body1"], joints["body2"], static, None if ids == "joint_index" else pid)
    assert sorted(order.tolist()) == list(range(nj))                  # a permutation
    assert offs[0] == 0 and offs[-1] == nj and (np.diff(offs) > 0).all()
    b1, b2 = joints["body1"].tolist(), joints["body2"].tolist()
    # units
    first = {}
    for j in range(nj - 1, -1, -1):
        first[int(pid[j])] = j
    partner = [-1] * nj
    for j in range(nj):
        if first[int(pid[j])] != j:
            continue
        o = first.get(int(pid[j]) ^ 1, -1)
        if o >= 0 and b1[o] == b1[j] and b2[o] == b2[j]:
            partner[j] = o
    follower = [partner[j] >= 0 and (int(pid[j]) & 1) == 1 for j in range(nj)]
    assert sum(follower) > 0 or name == "falling600" or (name == "synthetic_one_big_component" and ids == "joint_index")
    leaders = [j for j in range(nj) if not follower[j]]
    class_of = np.zeros(nj, dtype=np.int64)
    layouts = []
    for c in range(len(offs) - 1):
        sl = order[offs[c]:offs[c + 1]].tolist()
        class_of[sl] = c
        lead = [j for j in sl if not follower[j]]
        with_f = [j for j in lead if partner[j] >= 0]
        single = [j for j in lead if partner[j] < 0]
        layouts.append((sl, with_f, single))
        b = np.array([b1[j] for j in lead] + [b2[j] for j in lead])
        b = b[static[b] == 0]
        assert len(np.unique(b)) == len(b), "class %d: two units touch a dynamic body" % c
    # the colouring rule: two first-fit candidates over the UNITS in priority order — A = smallest free colour, B = two-ended
    # — and every connected component keeps the one that gives it fewer colours
    prio = {j: _priority(int(pid[j]), j, min(b1[j], b2[j])) for j in leaders}
    assert all(prio[j] == phyx_amd.schedule_priority(int(pid[j]), j, min(b1[j], b2[j])) for j in leaders[:200])      # (the library states the same rule)
    assert len(set(prio.values())) == len(leaders) and min(prio.values()) > 0
    parent = list(range(len(bodies)))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for j in range(nj):
        if not static[b1[j]] and not static[b2[j]]:
            parent[find(b1[j])] = find(b2[j])
    comp = {j: ("s", j) if static[b1[j]] and static[b2[j]] else ("c", find(b2[j] if static[b1[j]] else b1[j])) for j in leaders}
    degree = np.bincount(np.array([b1[j] for j in leaders] + [b2[j] for j in leaders]), minlength=len(bodies))      # units per body
    size = {}
    for j in leaders:
        size[comp[j]] = size.get(comp[j], 0) + (2 if partner[j] >= 0 else 1)          # joints of the component
    # a component of more than 1024 joints is partitioned: a unit with both bodies dynamic and in one block of 512 body indices
    # is INTERIOR AT LEVEL 0, otherwise — if they share a block of the same grid shifted by 256 — AT LEVEL 1; each level is a kind
    # of its own — own masks, own classes — and their classes come first in the group, level 0 before level 1
    nparts = (len(bodies) + 511) // 512

    def level_part(j):
        if not (comp[j][0] == "c" and size[comp[j]] > 1024) or static[b1[j]] or static[b2[j]]:
            return None
        if b1[j] // 512 == b2[j] // 512:
            return 0, b1[j] // 512
        if (b1[j] + 256) // 512 == (b2[j] + 256) // 512:
            return 1, nparts + (b1[j] + 256) // 512
        return None
    where = {j: level_part(j) for j in leaders}
    interior = {j: where[j] is not None for j in leaders}
    assert any(interior.values()) or name not in ("wall48x60", "synthetic_one_big_component")
    if name == "synthetic_one_big_component":
        kinds = [where[j][0] if where[j] else 2 for j in leaders]
        assert min(kinds.count(0), kinds.count(1), kinds.count(2)) > 50
    used_a, used_b, used_i, col_a, col_b, bad_b = {}, {}, ({}, {}), {}, {}, set()
    for j in sorted(leaders, key=lambda j: -int(prio[j])):
        dyn = [b for b in (b1[j], b2[j]) if not static[b]]
        if interior[j]:
            used = used_i[where[j][0]]
            mi = used.get(b1[j], 0) | used.get(b2[j], 0)
            ci = 0
            while mi >> ci & 1:
                ci += 1
            col_a[j] = ci
            col_b[j] = 0
            for b in dyn:
                used[b] = used.get(b, 0) | 1 << ci
            continue
        ma = 0
        mb = 0
        for b in dyn:
            ma |= used_a.get(b, 0)
            mb |= used_b.get(b, 0)
        ca = 0
        while ma >> ca & 1:
            ca += 1
        col_a[j] = ca
        k = min(max([int(degree[b]) for b in dyn] + [0]), 64)
        cb = None
        if min(b1[j], b2[j]) & 1:
            cb = next((c for c in range(k - 1, -1, -1) if not mb >> c & 1), None)
            if cb is None:
                cb = next((c for c in range(k, 64) if not mb >> c & 1), None)
        else:
            cb = next((c for c in range(64) if not mb >> c & 1), None)
        if cb is None:
            bad_b.add(comp[j])
            cb = 0
        else:
            for b in dyn:
                used_b[b] = used_b.get(b, 0) | 1 << cb
        col_b[j] = cb
        for b in dyn:
            used_a[b] = used_a.get(b, 0) | 1 << ca
    seen_a, seen_b = {}, {}
    ki0 = max([col_a[j] + 1 for j in leaders if interior[j] and where[j][0] == 0] + [0])      # the group's interior classes, per level
    ki = ki0 + max([col_a[j] + 1 for j in leaders if interior[j] and where[j][0] == 1] + [0])
    for j in leaders:
        if not interior[j]:
            seen_a.setdefault(comp[j], set()).add(col_a[j])
            seen_b.setdefault(comp[j], set()).add(col_b[j])
    for c, (sl, with_f, single) in enumerate(layouts):                # the layout of a class; an interior class is laid out part by part
        key = (lambda j: (where[j][1], j)) if c < ki else (lambda j: j)
        assert sl == sorted(with_f, key=key) + sorted(single, key=key) + [partner[j] for j in sorted(with_f, key=key)]
    for j in leaders:
        if interior[j]:
            assert class_of[j] == (ki0 if where[j][0] else 0) + col_a[j] < ki, "joint %d" % j
        else:
            use_b = comp[j] not in bad_b and size[comp[j]] <= 1024 and comp[j][0] == "c" and len(seen_b[comp[j]]) < len(seen_a[comp[j]])
            chosen, c = (seen_b[comp[j]], col_b[j]) if use_b else (seen_a[comp[j]], col_a[j])
            assert class_of[j] == ki + sum(1 for x in chosen if x < c), "joint %d" % j
        if partner[j] >= 0:
            assert class_of[partner[j]] == class_of[j]
    if name == "wall48x60":
        # what the partition buys: the interior classes hold most of the units and are one launch per sweep
        n_int = sum(1 for j in leaders if interior[j])
        assert n_int > 0.9 * len(leaders) and ki0 < ki < len(offs) - 1 <= ki + 4
    elif not ki:
        # and it never needs more classes than plain first-fit
        assert len(offs) - 1 <= max(col_a.values()) + 1


@pytest.mark.parametrize("name", list(SMALL_SCENES))
def test_island_partition_matches_oracle_gather(built_lib, oracle, name):
    """Same island partition as the GatherIslands restatement (ref: Solver.cpp:285-454), incl. coalescing."""
    import ctypes as C
    make, warm = SMALL_SCENES[name]
    bodies, _, joints = presolve_state(make(), warm)
    ji, sz = phyx_amd.schedule_islands(joints["body1"], joints["body2"], is_static(bodies))
    L = oracle.lib()
    nb, nj = len(bodies), len(joints)
    cap = nj + 64
    jidx = np.full(cap, -1, dtype=np.int32)
    off = np.zeros(nb + 1, dtype=np.int32)
    siz = np.zeros(nb + 1, dtype=np.int32)
    cnt, mx = C.c_int32(), C.c_int32()
    L.phxo_gather_islands(bodies.ctypes.data, nb, joints.ctypes.data, nj, 1, jidx.ctypes.data, cap,
                          off.ctypes.data, siz.ctypes.data, C.byref(cnt), C.byref(mx))
    assert cnt.value == len(sz) and list(siz[:cnt.value]) == list(sz)
    for i in range(cnt.value):
        members = jidx[off[i]:off[i] + siz[i]]
        assert (ji[members] == i).all()
        assert (np.diff(members) > 0).all()


def test_colouring_survives_compaction_of_the_joint_list(built_lib):
    """Island sharding solves a compacted subset of the joints.  With contactPointIndex as the priority id, a
    component's joints get the same colours whether or not other components' joints are in the list."""
    bodies, _, joints = presolve_state(scenes.stack(6, 30), 4)
    static = is_static(bodies)
    ji, _ = phyx_amd.schedule_islands(joints["body1"], joints["body2"], static)
    comp = np.where(static[joints["body1"]] == 0, joints["body1"], joints["body2"]) // 30      # column of the joint's dynamic body
    def colours(sel):
        sub = joints[sel]
        order, offs = phyx_amd.schedule_colours(sub["body1"], sub["body2"], static, sub["contact_point_index"])
        col = np.zeros(len(sub), dtype=np.int64)
        for c in range(len(offs) - 1):
            col[order[offs[c]:offs[c + 1]]] = c
        return col
    everything = colours(np.arange(len(joints)))
    keep = np.flatnonzero(comp % 2 == 0)                       # every other column
    assert 0 < len(keep) < len(joints)
    assert np.array_equal(colours(keep), everything[keep])


def test_schedule_handles_hub_and_empty(built_lib):
    # a dynamic hub touched by 200 joints needs 200 colours (> 64 exercises the multi-word masks)
    n = 200
    b1 = np.zeros(n, dtype=np.int32)
    b2 = np.arange(1, n + 1, dtype=np.int32)
    order, offs = phyx_amd.schedule_colours(b1, b2, np.zeros(n + 1, dtype=np.uint8))
    prio = [phyx_amd.schedule_priority(j, j, 0) for j in range(n)]
    assert len(offs) - 1 == n and list(order) == sorted(range(n), key=lambda j: -prio[j])      # one joint per colour, highest priority first
    # the same hub made static conflicts with nothing
    st = np.zeros(n + 1, dtype=np.uint8)
    st[0] = 1
    order, offs = phyx_amd.schedule_colours(b1, b2, st)
    assert len(offs) - 1 == 1
    order, offs = phyx_amd.schedule_colours([], [], [0, 0])
    assert len(order) == 0 and list(offs) == [0]
    with pytest.raises(phyx_amd.PhxError):
        phyx_amd.schedule_colours([0], [5], [0, 0])


def test_exchange_layout_partitions_the_groups_over_the_ranks():
    """The deal of the groups to the ranks and the segment layout of the island-sharded exchange (csrc/exchange.h, host-only):
    groups go longest-processing-time first by joint count (restated here), blocks never overlap, every rank's segment fits the
    common padded length, the joint load is balanced to within one group, and uniform groups are dealt round-robin."""
    import phyx_amd
    rng = np.random.default_rng(5)
    for ngroups, n in ((0, 1), (1, 8), (5, 2), (999, 8), (1000, 3), (37, 37), (12, 64)):
        gb = rng.integers(1, 769, ngroups).astype(np.int32)
        gs = rng.integers(1, 513, ngroups).astype(np.int32)
        off, rank_words, seg, owner = phyx_amd.exchange_layout(gb, gs, n)
        assert seg % 16384 == 0 and seg >= 8 and len(rank_words) == n          # padded to 64 KB (exchange.h XCH_SEGMENT_GRANULE_WORDS)
        assert int(rank_words.max()) <= seg < int(rank_words.max()) + 16384
        # longest processing time first, restated: decreasing joint count (ties: group number), to the least loaded rank (ties: lowest)
        load = [0] * n
        want = [0] * ngroups
        for g in sorted(range(ngroups), key=lambda g: (-int(gs[g]), g)):
            r = min(range(n), key=lambda r: (load[r], r))
            want[g] = r
            load[r] += int(gs[g])
        assert owner.tolist() == want
        if ngroups >= n:
            assert max(load) - min(load) <= int(gs.max())
        for r in range(n):
            mine = [g for g in range(ngroups) if owner[g] == r]
            at = 8                                                  # header words
            for g in mine:
                assert off[g] == at and off[g] % 4 == 0
                at += (6 * int(gb[g]) + 2 * int(gs[g]) + 3) // 4 * 4
            assert at == rank_words[r]
        # total payload is independent of the rank count
        assert int(rank_words.sum()) - 8 * n == int(sum((6 * int(b) + 2 * int(s) + 3) // 4 * 4 for b, s in zip(gb, gs)))
    _, _, _, owner = phyx_amd.exchange_layout([7] * 10, [440] * 10, 4)          # uniform columns: round-robin
    assert owner.tolist() == [g % 4 for g in range(10)]
    _, _, _, owner = phyx_amd.exchange_layout([9] * 5, [100, 100, 100, 100, 800], 2)       # one big group (an HBM group): alone on its rank
    assert owner.tolist() == [1, 1, 1, 1, 0]
    with pytest.raises(phyx_amd.PhxError):
        phyx_amd.exchange_layout([1], [1], 0)


BIN_CHUNK = 64          # csrc/schedule.h: a bin never spans a multiple of BIN_CHUNK component numbers


def _greedy_bins(sizes, units, cap_units):
    """The host loop of the schedule builder (csrc/solver_build.hip, step 3; csrc/schedule.hip): consecutive components packed greedily,
    every chunk of BIN_CHUNK component numbers starting a fresh bin."""
    bin_of, rank_of, goff = [-1] * len(sizes), [0] * len(sizes), [0]
    size = unit = rank = 0
    opened = False
    for c, (n, u) in enumerate(zip(sizes, units)):
        if c % BIN_CHUNK == 0:
            opened = False
        if n == 0:
            continue
        if not opened or size + n > 2 * cap_units or unit + u > cap_units:
            goff.append(goff[-1]); opened = True; size = unit = rank = 0
        bin_of[c] = len(goff) - 2; rank_of[c] = rank; rank += 1
        size += n; unit += u; goff[-1] += n
    return bin_of, rank_of, goff


def _chunk_bins(sizes, units, cap_units):
    """k_bin_components' formulation (csrc/schedule_kernels.h): a lane packs its chunk of BIN_CHUNK components on its own (pass 1:
    how many bins, how many slots), an exclusive scan over the lanes numbers the bins and places their slots, pass 2 writes."""
    n = len(sizes)
    chunks = [(c0, min(n, c0 + BIN_CHUNK)) for c0 in range(0, n, BIN_CHUNK)]

    def walk(c0, c1, first_bin, first_slot, out):
        size = unit = rank = 0
        opened, b, at, bins = False, first_bin - 1, first_slot, 0
        for c in range(c0, c1):
            if sizes[c] == 0:
                continue
            if not opened or size + sizes[c] > 2 * cap_units or unit + units[c] > cap_units:
                b += 1; bins += 1; opened = True; size = unit = rank = 0
                if out is not None:
                    out[2][b] = at
            if out is not None:
                out[0][c] = b; out[1][c] = rank
            rank += 1; size += sizes[c]; unit += units[c]; at += sizes[c]
        return bins, at - first_slot
    counts = [walk(c0, c1, 0, 0, None) for c0, c1 in chunks]
    nbins, nslots = sum(b for b, _ in counts), sum(s for _, s in counts)
    out = ([-1] * n, [0] * n, [0] * (nbins + 1))
    b = s = 0
    for (c0, c1), (cb, cs) in zip(chunks, counts):
        walk(c0, c1, b, s, out)
        b += cb; s += cs
    out[2][nbins] = nslots
    return out


def test_binning_by_chunks_equals_the_greedy_loop():
    """The device's binning (a lane per chunk of 64 components, a scan over the lanes) against the sequential loop of the host
    builders, on random component sizes incl. empty components, components that fill a bin alone and long runs of tiny ones;
    and what the chunk rule costs in bins against unbroken greedy packing."""
    rng = np.random.default_rng(7)
    for case in range(300):
        n = int(rng.integers(1, 700))
        cap = int(rng.choice([256, 512]))
        kind = case % 4
        if kind == 0:
            sizes = rng.integers(0, 2 * cap + 1, size=n)
        elif kind == 1:
            sizes = rng.integers(0, 40, size=n)
        elif kind == 2:
            sizes = np.where(rng.random(n) < 0.5, 0, rng.integers(1, 2 * cap + 1, size=n))
        else:
            sizes = np.full(n, 440 if cap == 256 else 1000)
        units = np.minimum(cap, (sizes + 1) // 2 + rng.integers(0, 3, size=n) * (sizes > 0))
        units = np.where(sizes > 0, np.maximum(units, (sizes + 1) // 2), 0)
        units = np.minimum(units, np.minimum(sizes, cap))
        g = _greedy_bins(list(map(int, sizes)), list(map(int, units)), cap)
        c = _chunk_bins(list(map(int, sizes)), list(map(int, units)), cap)
        assert g[0] == c[0] and g[1] == c[1] and g[2] == c[2], (case, n, cap)
    # the price of the chunk boundaries: the 1e4 columns of the 1M-box world (206-joint components, two to a bin)
    sizes, units = [206] * 10000, [103] * 10000
    assert len(_greedy_bins(sizes, units, 256)[2]) - 1 == 5000


def _layout_classes(units, T):
    """csrc/schedule.h layout_classes, restated: classes in order; one that would straddle a wave more than its size needs starts on
    the next wave boundary if everything behind it still fits; a small class goes into the gap such a move left."""
    remaining, cursor, gap_at, gap_n, begin = sum(units), 0, 0, 0, []
    for n in units:
        if n <= gap_n:
            begin.append(gap_at); gap_at += n; gap_n -= n
        else:
            at = cursor
            aligned = (at + 63) & ~63
            if (at & 63) + n > ((n + 63) & ~63) and aligned + remaining <= T:
                gap_at, gap_n, at = cursor, aligned - cursor, aligned
            begin.append(at); cursor = at + n
        remaining -= n
    return begin


def _wave_passes(begin, units):
    return sum((b + n - 1) // 64 - b // 64 + 1 for b, n in zip(begin, units) if n)


@pytest.mark.parametrize("name,lanes,body_cap", [("stack10x100", 256, 768), ("stack2x10", 256, 768), ("falling600", 256, 768), ("stack3x500", 512, 1024)])
def test_island_groups_and_their_lanes(built_lib, name, lanes, body_cap):
    """phx_schedule_groups (host-only): the island-mode schedule as workgroup-sized groups — groups are body-disjoint and respect the
    caps, every group's classes are body-disjoint sets of units in the layout of a class, and the LANES the island kernel gives the
    units (csrc/schedule.h LANES, restated above): one lane per unit, a class's units on consecutive lanes in slot order, the classes'
    ranges where layout_classes puts them — never more wave passes per sweep than back-to-back ranges, and 6 instead of 7 for the
    200-box column of BASELINE config 2."""
    if name == "stack3x500":
        bodies, _, joints = presolve_state(scenes.stack(3, 500), 3, iters=30)
    else:
        make, warm = SMALL_SCENES[name]
        bodies, _, joints = presolve_state(make(), warm)
    static = is_static(bodies)
    nj = len(joints)
    b1, b2 = joints["body1"], joints["body2"]
    r = phyx_amd.schedule_groups(b1, b2, static, joints["contact_point_index"], lanes=lanes, body_cap=body_cap)
    order, offs, goff, gfc, lg = r["order"], r["colour_offsets"], r["group_offsets"], r["group_first_colour"], r["lds_groups"]
    assert sorted(order.tolist()) == list(range(nj)) and offs[0] == 0 and offs[-1] == nj and goff[0] == 0 and goff[-1] == nj
    assert lg >= 1 and len(goff) in (lg + 1, lg + 2)
    owner = {}
    lane_of = dict(zip(r["unit_leader_slot"].tolist(), r["unit_lane"].tolist()))
    seen_units = 0
    for g in range(lg):
        slots = range(goff[g], goff[g + 1])
        touched = set()
        for s in slots:
            for body in (int(b1[order[s]]), int(b2[order[s]])):
                touched.add(body)
                if not static[body]:
                    assert owner.setdefault(body, g) == g, "a dynamic body in two groups"
        assert len(slots) <= 2 * lanes and len(touched) <= body_cap
        units, lanes_of_class = [], []
        for c in range(gfc[g], gfc[g + 1]):
            sl = list(range(offs[c], offs[c + 1]))
            leaders = [s for s in sl if s in lane_of]
            assert leaders == sl[:len(leaders)] and len(leaders) >= len(sl) - len(leaders)      # leaders first, then the followers
            dyn = [b for s in leaders for b in (int(b1[order[s]]), int(b2[order[s]])) if not static[b]]
            assert len(dyn) == len(set(dyn)), "two units of a class share a dynamic body"
            units.append(len(leaders))
            lanes_of_class.append([lane_of[s] for s in leaders])
        seen_units += sum(units)
        begin = _layout_classes(units, lanes)
        flat = [l for ls in lanes_of_class for l in ls]
        assert len(set(flat)) == len(flat) and max(flat) < lanes and sum(units) <= lanes
        for b, ls in zip(begin, lanes_of_class):
            assert ls == list(range(b, b + len(ls)))
        plain = [sum(units[:c]) for c in range(len(units))]
        assert _wave_passes(begin, units) <= _wave_passes(plain, units)
    assert seen_units == len(lane_of)
    assert _layout_classes([100, 96, 5, 4], 256) == [0, 128, 100, 105]
    assert _wave_passes([0, 128, 100, 105], [100, 96, 5, 4]) == 6 and _wave_passes([0, 100, 196, 201], [100, 96, 5, 4]) == 7
    assert _layout_classes([250, 246, 8, 6], 512) == [0, 250, 496, 504]                      # (a 500-box column fills its 512 lanes: nothing to move)
    assert _layout_classes([250, 240, 8, 6], 512) == [0, 256, 496, 250]                      # (the gap of 6 lanes takes the class that fits it)
