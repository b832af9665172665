"""Deterministic scene generators for the headless bench and the parity tests.

The reference only ships interactive demo scenes (ref: src/main.cpp:82-230).  `stack` keeps the
box size / spacing of its scene 4 ("Stacks", main.cpp:154-168) without the taper so boxes rest
exactly on each other; `falling` keeps the shape of scene 0 (main.cpp:97-110) but draws from
splitmix64 instead of libc rand() so the scene is reproducible everywhere.  A scene is a dict of
equal-length numpy arrays: px, py, angle, sx, sy (half sizes), static (bool).
"""
import numpy as np


def _scene(px, py, angle, sx, sy, static):
    return {
        "px": np.asarray(px, dtype=np.float32), "py": np.asarray(py, dtype=np.float32),
        "angle": np.asarray(angle, dtype=np.float32), "sx": np.asarray(sx, dtype=np.float32),
        "sy": np.asarray(sy, dtype=np.float32), "static": np.asarray(static, dtype=bool),
    }


def stack(nx, ny, x_offset_columns=0):
    """Ground (body 0, static, half-size (max(nx,1)*15, 10) at the origin) + nx columns of ny boxes of
    half-size 5x5, column pitch 15, row pitch 10.  `x_offset_columns` shifts the
    columns sideways (used to give each rank of a multi-GPU run its own slab of one wide world)."""
    n = nx * ny
    xs = np.repeat(np.arange(nx, dtype=np.int64), ny)
    ys = np.tile(np.arange(ny, dtype=np.int64), nx)
    px = np.concatenate([[0.0 + x_offset_columns * 15.0], (xs - nx // 2 + x_offset_columns) * 15.0]).astype(np.float32)
    py = np.concatenate([[0.0], 15.0 + 10.0 * ys]).astype(np.float32)
    sx = np.concatenate([[max(nx, 1) * 15.0], np.full(n, 5.0)]).astype(np.float32)
    sy = np.concatenate([[10.0], np.full(n, 5.0)]).astype(np.float32)
    static = np.zeros(n + 1, dtype=bool)
    static[0] = True
    return _scene(px, py, np.zeros(n + 1), sx, sy, static)


def wall(nx, ny, pitch=10.5):
    """Ground + ny rows of nx boxes of half-size 5x5 laid like bricks (odd rows shifted by half a pitch), bodies numbered row by
    row: every box rests on two boxes of the row below, so the whole wall is ONE connected component — the shape of island the
    HBM path takes (a settled pile), at a size the CPU oracle can still replay (48 x 60: ~1.1e4 joints)."""
    n = nx * ny
    r = np.repeat(np.arange(ny, dtype=np.int64), nx)
    i = np.tile(np.arange(nx, dtype=np.int64), ny)
    px = np.concatenate([[0.0], (i - nx // 2) * pitch + (r & 1) * 0.5 * pitch]).astype(np.float32)
    py = np.concatenate([[0.0], 15.0 + 10.0 * r]).astype(np.float32)
    sx = np.concatenate([[max(nx, 1) * pitch], np.full(n, 5.0)]).astype(np.float32)
    sy = np.concatenate([[10.0], np.full(n, 5.0)]).astype(np.float32)
    static = np.zeros(n + 1, dtype=bool)
    static[0] = True
    return _scene(px, py, np.zeros(n + 1), sx, sy, static)


def clique(n, pitch=0.01):
    """Ground + n boxes of half-size 5x5 dropped almost on top of each other: every box overlaps every other one, so each
    body carries ~2n joints.  More than 64 colours are needed — the case where the device schedule builder hands over to
    the host builder — and the narrowphase merges contact points all the time."""
    px = np.concatenate([[0.0], pitch * np.arange(n)]).astype(np.float32)
    py = np.concatenate([[0.0], np.full(n, 14.0)]).astype(np.float32)
    sx = np.concatenate([[100.0], np.full(n, 5.0)]).astype(np.float32)
    sy = np.concatenate([[10.0], np.full(n, 5.0)]).astype(np.float32)
    static = np.zeros(n + 1, dtype=bool)
    static[0] = True
    return _scene(px, py, np.zeros(n + 1), sx, sy, static)


def _splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def falling(n, seed=1, half=4.0, width=500.0, ymin=50.0, ymax=1000.0, ground_half_width=10000.0):
    """Ground + n boxes of half-size `half` scattered uniformly (reference scene 0 shape)."""
    s = seed
    px, py = [0.0], [0.0]
    for _ in range(n):
        s, a = _splitmix64(s)
        s, b = _splitmix64(s)
        u = (a >> 40) / float(1 << 24)
        v = (b >> 40) / float(1 << 24)
        px.append(-width + 2.0 * width * u)
        py.append(ymin + (ymax - ymin) * v)
    sx = [ground_half_width] + [half] * n
    sy = [10.0] + [half] * n
    static = [True] + [False] * n
    return _scene(px, py, np.zeros(n + 1), sx, sy, static)


def piles(clusters, per_cluster, pitch=110.0, spread=25.0, seed=11, ymax=600.0):
    """Ground + `clusters` groups of `per_cluster` falling boxes, the groups `pitch` apart and each scattered over +-`spread`:
    separate piles that widen as they settle and grow into their neighbours — islands that wander across any fixed cut of the
    x axis (the re-slab case of an ownership-sharded world, phyx_amd.dist.SlabWorld)."""
    px, py = [0.0], [0.0]
    for k in range(clusters):
        sc = falling(per_cluster, seed=seed + k, half=4.0, width=spread, ymin=40.0, ymax=ymax)
        px += list(sc["px"][1:] + np.float32((k - (clusters - 1) / 2.0) * pitch))
        py += list(sc["py"][1:])
    n = clusters * per_cluster
    return _scene(px, py, np.zeros(n + 1), [clusters * pitch + 1000.0] + [4.0] * n, [10.0] + [4.0] * n, [True] + [False] * n)


def tilted(n, seed=7):
    """Small scene of rotated boxes dropped on the ground — exercises the vertex/edge contact cases
    (ref: Collider.cpp:94-209) that axis-aligned stacks never reach."""
    sc = falling(n, seed=seed, half=6.0, width=60.0, ymin=20.0, ymax=200.0, ground_half_width=400.0)
    s = seed * 977
    ang = [0.0]
    for _ in range(n):
        s, a = _splitmix64(s)
        ang.append(-1.5 + 3.0 * ((a >> 40) / float(1 << 24)))
    sc["angle"] = np.asarray(ang, dtype=np.float32)
    return sc


def reference(scene, boxes=400, seed=3):
    """Headless, scaled-down versions of the reference's demo scenes (ref: src/main.cpp:82-230; libc rand() replaced by
    splitmix64): every scene starts with the ground (static, half-size 10000 x 10) and the 30 x 30 box at (-1000, 1500)
    (main.cpp:90-95).  `boxes` scales the body count.
      0 'Falling'  2 'Pyramid'  3 'Reverse Pyramid'  4 'Stacks' (tapered boxes)  5 shelves + falling
      6 'Dual Stacks' (a tilted static plank)  7 'Islands' (static splitters)
    Scene 1 ('Wall', boxes touching sideways) is left out: it overflows base/DenseHash.h:191 in the reference itself
    (SURVEY.md §8c).  Scenes 5 and 6 pin their shelves with invMass = 0 ONLY (main.cpp:176-177, 194-195): key 'pinned'."""
    px, py, ang, sx, sy, static, pinned = [0.0, -1000.0], [0.0, 1500.0], [0.0, 0.0], [10000.0, 30.0], [10.0, 30.0], [True, False], [False, False]
    state = [seed]

    def rnd(lo, hi):
        state[0], z = _splitmix64(state[0])
        return lo + (hi - lo) * ((z >> 40) / float(1 << 24))

    def add(x, y, hx, hy, angle=0.0, fixed=False, pin=False):
        px.append(x); py.append(y); ang.append(angle); sx.append(hx); sy.append(hy); static.append(fixed); pinned.append(pin)

    k = scene % 8
    if k == 0:
        for _ in range(boxes):
            add(rnd(-500.0, 500.0) * 0.15, rnd(50.0, 1000.0) * 0.3, 4.0, 4.0)
    elif k == 2:
        n = max(4, boxes // 8)
        for step in range(n):
            add(0.0, 15.0 + (n - 1 - step) * 10.0, 10.0 + step * 5.0, 5.0)        # widest at the bottom, resting (the demo drops it from y = 1005)
    elif k == 3:
        n = max(4, boxes // 8)
        for step in range(n):
            add(0.0, 15.0 + step * 10.0, 10.0 + step * 5.0, 5.0)
    elif k == 4:
        cols = max(2, boxes // 40)
        for left in range(-(cols // 2), cols - cols // 2):
            for b in range(40):
                add(left * 15.0, 15.0 + b * 10.0, 5.0 - b * 0.03, 5.0)
    elif k == 5:
        add(0.0, 400.0, 600.0, 10.0, pin=True)
        add(800.0, 200.0, 400.0, 10.0, pin=True)
        for _ in range(boxes):
            add(rnd(0.0, 500.0) * 0.4, 415.0 + rnd(0.0, 2000.0) * 0.1, 4.0, 4.0)
    elif k == 6:
        add(0.0, 400.0, 600.0, 10.0, pin=True)
        add(800.0, 200.0, 400.0, 10.0, pin=True)
        add(500.0, 500.0, 600.0, 10.0, angle=-0.5, fixed=True)
        for _ in range(boxes // 2):
            add(200.0 + rnd(0.0, 300.0) * 0.3, 700.0 + rnd(0.0, 2000.0) * 0.08, 4.0, 4.0)
            add(-500.0 + rnd(0.0, 300.0) * 0.3, 415.0 + rnd(0.0, 2000.0) * 0.08, 4.0, 4.0)
    elif k == 7:
        groups = 3
        for g in range(-(groups // 2), groups - groups // 2):
            add(g * 300.0, 500.0, 20.0, 1000.0, fixed=True)
            for _ in range(boxes // groups):
                add(g * 300.0 + rnd(50.0, 250.0) * 0.5, rnd(50.0, 1500.0) * 0.12, 4.0, 4.0)
    else:
        raise ValueError("scene 1 ('Wall') is not reproduced: it overflows the reference's own DenseHash")
    sc = _scene(px, py, ang, sx, sy, static)
    sc["pinned"] = np.asarray(pinned, dtype=bool)
    return sc
