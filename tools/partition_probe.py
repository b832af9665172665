"""Would a partitioned solve of the settled world's merged island pay?  

The 200k-box stack after `steps` world steps is one island of ~8e5 joints that the HBM path sweeps class by class, one launch
each.  Parts = blocks of `B` consecutive body indices; a unit is INTERIOR if its dynamic bodies lie in one part and it touches no
static body, everything else is INTERFACE.  Interior classes of all parts could run in ONE launch per sweep (a workgroup per part,
bodies in LDS); interface classes stay one launch each.  Prints, per B: the split, the classes either side needs (first-fit in
the product's priority order is approximated by a random order), and the launches per sweep against today's.
usage: partition_probe.py [steps=57] [columns=1000] [rows=200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 57
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 200
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(cols, rows))
cfg = Configuration(2, 2, 20, 20)
for _ in range(steps): w.Update(1 / 60, cfg)
st = w.solver_stats() if hasattr(w, "solver_stats") else None
b, j = w.bodies, w.contactJoints
print("bodies %d joints %d" % (len(b), len(j)))
static = (b["inv_mass"] == 0) & (b["inv_inertia"] == 0)
cp = j["contact_point_index"].astype(np.int64); b1 = j["body1"].astype(np.int64); b2 = j["body2"].astype(np.int64)
man = cp // 2
_, first = np.unique(man, return_index=True)            # one unit per manifold
u1, u2 = b1[first], b2[first]
nu = len(first)
print("units %d; units touching a static body %d" % (nu, int((static[u1] | static[u2]).sum())))
rng = np.random.default_rng(1)


def first_fit(a, c, n_bodies, order):
    """classes by first fit over `order`; a, c = the unit's dynamic bodies (-1 = none)"""
    used = {}
    ncls = 0
    cls = np.zeros(len(a), np.int32)
    for u in order:
        m = 0
        x, y = a[u], c[u]
        if x >= 0: m |= used.get(x, 0)
        if y >= 0: m |= used.get(y, 0)
        k = 0
        while m >> k & 1: k += 1
        cls[u] = k
        if x >= 0: used[x] = used.get(x, 0) | (1 << k)
        if y >= 0: used[y] = used.get(y, 0) | (1 << k)
        if k + 1 > ncls: ncls = k + 1
    return ncls, cls


d1 = np.where(static[u1], -1, u1); d2 = np.where(static[u2], -1, u2)
order = rng.permutation(nu)
total, _ = first_fit(d1, d2, len(b), order)
print("whole island, first fit in random order: %d classes (= launches per sweep today)" % total)
for B in (200, 256, 400, 512, 768, 1024, 2048):
    for off in (1, 0):
        p1 = np.where(d1 >= 0, (d1 - off) // B, -1); p2 = np.where(d2 >= 0, (d2 - off) // B, -1)
        interior = (p1 >= 0) & (p2 >= 0) & (p1 == p2)
        ni = int(interior.sum())
        oi = order[interior[order]]
        ki, _ = first_fit(d1, d2, len(b), oi)          # parts are body-disjoint: one pass colours them all, max = classes of the worst part
        oe = order[~interior[order]]
        ke, ce = first_fit(d1, d2, len(b), oe)
        sizes = np.bincount(ce[~interior], minlength=ke)
        print("B=%5d off=%d: interior %7d (%.1f %%) in %d classes -> 1 launch; interface %7d in %d classes (sizes %s) -> %d launches per sweep"
              % (B, off, ni, 100.0 * ni / nu, ki, nu - ni, ke, " ".join(str(int(x)) for x in sizes), 1 + ke))
