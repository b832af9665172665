"""Generate tests/golden/leaf_vectors.npz from the REAL reference headers.

Runs only where /root/reference exists (the build container): it calls oracle/_ref/libphyx_ref_leaf.so,
which oracle/Makefile compiles straight from /root/reference/src headers (no stand-ins, no copies).
The .npz holds data only — inputs and the reference's outputs — and travels with the repo, so the
CPU suite can pin the oracle's leaf functions on machines that have neither the reference nor the shim.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402


def main():
    R = ob.ref_lib()
    if R is None:
        raise SystemExit("oracle/_ref/libphyx_ref_leaf.so is absent: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(20260928)
    out = {}
    out["sizes"] = np.array([R.ref_sizeof(i) for i in range(19)], dtype=np.int32)
    out["enums"] = np.array([R.ref_config_enum(i) for i in range(7)], dtype=np.int32)

    # radixFloat (ref: base/RadixSort.h:19-26)
    f = np.concatenate([
        (rng.standard_normal(4000) * 10.0 ** rng.integers(-30, 30, 4000)).astype(np.float32),
        np.array([0.0, -0.0, 1e-45, -1e-45, 1.17549435e-38, -1.17549435e-38, 3.4028235e38, -3.4028235e38,
                  np.inf, -np.inf, 1.0, -1.0, 5.0, -5.0, 7492.5, -7507.5], dtype=np.float32)])
    out["radix_float_in"] = f
    out["radix_float_out"] = np.array([R.ref_radix_float(float(v)) for v in f], dtype=np.uint32)

    # radixSort3 (ref: base/RadixSort.h:28-95): heavy ties, ordered and random keys
    n = 20000
    keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    keys[:3000] = keys[0]
    keys[3000:6000] = (keys[3000:6000] & np.uint32(0x7FF)) | np.uint32(0x80000000)
    e = np.zeros(n, dtype=ob.sort_entry_dtype)
    e["value"] = keys
    e["index"] = np.arange(n, dtype=np.uint32)
    out["sort_in"] = e.copy()
    scratch = np.zeros_like(e)
    R.ref_radix_sort3(e.ctypes.data, scratch.ctypes.data, n)
    out["sort_out"] = e

    # pair hash (ref: Collider.h:10-18)
    hp = rng.integers(0, 2 ** 32, (2000, 2), dtype=np.uint64).astype(np.uint32)
    out["hash_in"] = hp
    out["hash_out"] = np.array([R.ref_pair_hash(int(a), int(b)) for a, b in hp], dtype=np.uint32)

    # RigidBody construction + RecomputeAABB + Rotate + GetSupportPointSet (ref: RigidBody.h:15-41,
    # Coords2.h:10-17, Geom.h:66-85, Vector2.h:48-56)
    m = 1500
    args = np.zeros((m, 6), dtype=np.float32)
    args[:, 0] = rng.uniform(-1e4, 1e4, m)
    args[:, 1] = rng.uniform(-1e3, 1e4, m)
    args[:, 2] = rng.uniform(-3.2, 3.2, m)
    args[:200, 2] = 0.0
    args[:, 3] = rng.uniform(0.5, 60.0, m)
    args[:, 4] = rng.uniform(0.5, 60.0, m)
    args[:, 5] = 1e-5
    bodies = np.zeros(m, dtype=ob.body_dtype)
    rot_angle = rng.uniform(-0.4, 0.4, m).astype(np.float32)
    rotated = np.zeros(m, dtype=ob.body_dtype)
    axes = rng.standard_normal((m, 2)).astype(np.float32)
    sup_n = np.zeros(m, dtype=np.int32)
    sup_pts = np.zeros((m, 4), dtype=np.float32)
    for k in range(m):
        one = np.zeros(1, dtype=ob.body_dtype)
        R.ref_body_init(one.ctypes.data, *[float(x) for x in args[k]])
        bodies[k] = one[0]
        if k % 3 == 0:      # axis aligned with the body frame -> the edge (2-point) case
            axes[k] = (one["xv"]["x"][0], one["xv"]["y"][0])
        elif k % 3 == 1:
            axes[k] = (-one["yv"]["x"][0], -one["yv"]["y"][0])
        tmp = np.zeros(4, dtype=np.float32)
        sup_n[k] = R.ref_support_points(one.ctypes.data, float(axes[k, 0]), float(axes[k, 1]), tmp.ctypes.data)
        sup_pts[k] = tmp
        R.ref_coords_rotate(one.ctypes.data, float(rot_angle[k]))
        R.ref_update_geom(one.ctypes.data)
        rotated[k] = one[0]
    out["body_args"] = args
    out["body_init"] = bodies
    out["rot_angle"] = rot_angle
    out["body_rotated"] = rotated
    out["support_axis"] = axes
    out["support_n"] = sup_n
    out["support_pts"] = sup_pts

    # narrowphase leaves that live in headers: ContactPoint ctor + Equals (ref: Manifold.h:18-36), the plane form of
    # ProjectPointToLine (ref: Vector2.h:277-285), AABB2::Intersects (ref: AABB2.h:18-24)
    q = 1200
    ia, ib = rng.integers(0, m, q), rng.integers(0, m, q)
    cp_args = rng.uniform(-50, 50, (q, 6)).astype(np.float32)
    cp_args[:, 0:2] += np.stack([bodies["pos"]["x"][ia], bodies["pos"]["y"][ia]], axis=1)
    cp_args[:, 2:4] += np.stack([bodies["pos"]["x"][ib], bodies["pos"]["y"][ib]], axis=1)
    made = np.zeros(q, dtype=ob.contact_point_dtype)
    for k in range(q):
        one = np.zeros(1, dtype=ob.contact_point_dtype)
        R.ref_contact_point_make(one.ctypes.data, *[float(x) for x in cp_args[k]], bodies[ia[k]:ia[k] + 1].ctypes.data, bodies[ib[k]:ib[k] + 1].ctypes.data)
        made[k] = one[0]
    out["cp_args"], out["cp_b1"], out["cp_b2"], out["cp_made"] = cp_args, ia.astype(np.int32), ib.astype(np.int32), made
    # Equals: pairs at distances straddling the tolerance on either delta
    ea = made.copy()
    eb = made.copy()
    jitter = rng.uniform(-3.0, 3.0, (q, 4)).astype(np.float32)
    jitter[: q // 3, 2:] = 0.0                      # only delta1 differs
    jitter[q // 3: 2 * q // 3, :2] = 0.0            # only delta2 differs
    eb["delta1"]["x"] += jitter[:, 0]; eb["delta1"]["y"] += jitter[:, 1]; eb["delta2"]["x"] += jitter[:, 2]; eb["delta2"]["y"] += jitter[:, 3]
    tol = np.where(np.arange(q) % 2 == 0, 2.0, 0.75).astype(np.float32)
    out["eq_a"], out["eq_b"], out["eq_tol"] = ea, eb, tol
    out["eq_out"] = np.array([R.ref_contact_equals(ea[k:k + 1].ctypes.data, eb[k:k + 1].ctypes.data, float(tol[k])) for k in range(q)], dtype=np.int32)
    pl = rng.uniform(-100, 100, (q, 8)).astype(np.float32)
    pl[:, 4:6] /= np.linalg.norm(pl[:, 4:6], axis=1, keepdims=True)
    pl[:, 6:8] = pl[:, 4:6] * rng.choice([-1.0, 1.0], (q, 1)).astype(np.float32)   # narrowphase projects along +-normal
    pl[: q // 4, 6:8] = rng.standard_normal((q // 4, 2)).astype(np.float32)         # and a general direction
    pr = np.zeros((q, 2), dtype=np.float32)
    for k in range(q):
        tmp = np.zeros(2, dtype=np.float32)
        R.ref_project_point_to_line(*[float(x) for x in pl[k]], tmp.ctypes.data)
        pr[k] = tmp
    out["proj_in"], out["proj_out"] = pl, pr
    # AABB overlap, including touching boxes (shared edges are overlaps: the tests are strict '>')
    ta, tb = rng.integers(0, m, q), rng.integers(0, m, q)
    touch = bodies.copy()
    out["aabb_a"], out["aabb_b"] = ta.astype(np.int32), tb.astype(np.int32)
    out["aabb_out"] = np.array([R.ref_aabb_intersects(touch[a:a + 1].ctypes.data, touch[b:b + 1].ctypes.data) for a, b in zip(ta, tb)], dtype=np.int32)
    edge = np.zeros(4, dtype=ob.body_dtype)
    for k, (x, y) in enumerate([(0, 0), (20, 0), (0, 20), (20.5, 0)]):
        one = np.zeros(1, dtype=ob.body_dtype)
        R.ref_body_init(one.ctypes.data, float(x), float(y), 0.0, 10.0, 10.0, 1e-5)
        edge[k] = one[0]
    out["aabb_edge_bodies"] = edge
    out["aabb_edge_out"] = np.array([[R.ref_aabb_intersects(edge[a:a + 1].ctypes.data, edge[b:b + 1].ctypes.data) for b in range(4)] for a in range(4)], dtype=np.int32)

    # DenseHashSet insert-only behaviour == set semantics (ref: base/DenseHash.h:208-236)
    ps = rng.integers(0, 300, (5000, 2), dtype=np.uint64).astype(np.uint32)
    res = np.zeros(len(ps), dtype=np.uint8)
    R.ref_pairset_insert_run(ps.ctypes.data, len(ps), res.ctypes.data)
    out["pairset_in"] = ps
    out["pairset_new"] = res

    # scalar SIMD wrapper semantics (ref: base/SIMD_Scalar.h:265-278)
    fx = np.array([1.5, -2.0, 0.0, -0.0, 3.0, 1e-4], dtype=np.float32)
    fy = np.array([-1.0, 2.0, -0.0, 0.0, -0.0, -1e-9], dtype=np.float32)
    out["flipsign_x"], out["flipsign_y"] = fx, fy
    out["flipsign_out"] = np.array([R.ref_flipsign1(float(a), float(b)) for a, b in zip(fx, fy)], dtype=np.float32)
    out["max_out"] = np.array([R.ref_max1(float(a), float(b)) for a, b in zip(fx, fy)], dtype=np.float32)
    # the SSE2 / AVX2 wrappers of flipsign / max / abs (ref: base/SIMD_SSE2.h, base/SIMD_AVX2.h:260-285): signed zeros,
    # denormals, infinities; no NaNs (the solve loops never feed one to max)
    sx = np.array([1.5, -2.0, 0.0, -0.0, 3.0, 1e-4, 1e-40, -1e-40, np.inf, -np.inf, 0.3, -0.3, 7.0, -7.0], dtype=np.float32)
    sy = np.array([-1.0, 2.0, -0.0, 0.0, -0.0, -1e-9, -1e-45, 1e-45, -np.inf, 2.0, 0.0, -0.0, 7.0, -8.0], dtype=np.float32)
    out["simd_x"], out["simd_y"] = sx, sy
    for width, fn in ((4, R.ref_simd4_lane0), (8, R.ref_simd8_lane0)):
        for op, name in enumerate(("flipsign", "max", "abs")):
            out["simd%d_%s" % (width, name)] = np.array([fn(op, float(a), float(b)) for a, b in zip(sx, sy)], dtype=np.float32)

    # ContactJointPacked<N> field offsets in words, N = 1, 4, 8 (ref: Solver.h:7-45)
    out["packed_offsets"] = np.array([[R.ref_packed_offset(n, f) for f in range(13)] for n in (1, 4, 8)], dtype=np.int32)

    # the AVX2 gather / scatter of eight 16-byte SolveBody records (ref: base/SIMD_AVX2.h:324-377)
    import ctypes as C
    recs = rng.standard_normal((64, 4)).astype(np.float32)
    base = np.zeros(64 * 4 + 16, dtype=np.float32)
    off = (-base.ctypes.data // 4) % 4                          # 16-byte aligned start inside the buffer
    arr = base[off:off + 256].reshape(64, 4)
    arr[:] = recs
    idx = rng.permutation(64)[:8 * 6].astype(np.int32).reshape(6, 8)
    g = np.zeros((6, 32), dtype=np.float32)
    for k in range(6):
        R.ref_loadindexed4_v8(C.c_void_p(arr.ctypes.data), idx[k].ctypes.data_as(C.c_void_p), 16, g[k].ctypes.data_as(C.c_void_p))
    out["gather_records"], out["gather_indices"], out["gather_out"] = recs, idx, g
    lanes = rng.standard_normal((6, 32)).astype(np.float32)
    sc = np.zeros((6, 64, 4), dtype=np.float32)
    for k in range(6):
        arr[:] = 0
        R.ref_storeindexed4_v8(lanes[k].ctypes.data_as(C.c_void_p), C.c_void_p(arr.ctypes.data), idx[k].ctypes.data_as(C.c_void_p), 16)
        sc[k] = arr
    out["scatter_lanes"], out["scatter_out"] = lanes, sc

    # DenseHashSet with erases, on a run where no inserted key is already present (ref: base/DenseHash.h:208-236)
    keys = rng.permutation(4000)[:3000].astype(np.uint32)
    pairs = np.stack([keys // 61, keys % 61 + 100], axis=1).astype(np.uint32)           # 3000 distinct pairs
    seq, ops = [], []
    for k in range(1500): seq.append(pairs[k]); ops.append(0)
    for k in range(0, 1500, 3): seq.append(pairs[k]); ops.append(1)                        # erase every third
    for k in range(1500, 3000): seq.append(pairs[k]); ops.append(0)                        # fresh keys over the tombstones
    for k in range(1, 1500, 7): seq.append(pairs[k]); ops.append(1)
    seq = np.array(seq, dtype=np.uint32); ops = np.array(ops, dtype=np.uint8)
    member = np.zeros(len(seq), dtype=np.uint8)
    R.ref_pairset_mixed_run.restype = C.c_size_t
    size = R.ref_pairset_mixed_run(seq.ctypes.data_as(C.c_void_p), ops.ctypes.data_as(C.c_void_p), C.c_size_t(len(seq)), member.ctypes.data_as(C.c_void_p))
    out["pairset_mixed_pairs"], out["pairset_mixed_ops"], out["pairset_mixed_member"] = seq, ops, member
    out["pairset_mixed_size"] = np.array([size], dtype=np.int64)

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "leaf_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
