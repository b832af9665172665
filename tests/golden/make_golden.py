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

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "leaf_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
