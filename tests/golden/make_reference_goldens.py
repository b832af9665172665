"""Generate tests/golden/reference_*.npz from the REAL reference World / Collider / Solver.

Needs oracle/_ref/libphyx_ref_full_{fast,strict}.so, which `make -C oracle ref_full` builds from /root/reference/src as it
lies — possible only once the reference's microprofile submodule is present (src/microprofile/microprofile.h); until then
this script exits with a message and tests/test_reference_goldens.py skips.  The .npz files hold data only: inputs and the
reference's outputs.  workers = 0 throughout (bit-identical run to run).

Per scene (2x50 stack, 10x100 stack, a 1k 'falling' scene with this repo's PRNG) and per step s in {1,2,3}:
  solver inputs     bodies (raw 128-B records), contact points, contact joints (warm-start impulses included)
  ordering          joint_index after PrepareIndices for N = 1, 4, 8, island offsets / sizes (Multiple mode)
  refresh           ContactJointPacked<N> blocks after a 0-iteration solve
  per iteration     bodies' four velocity fields + joint impulses after k = 0..20 impulse iterations (re-run from the same state)
  outputs           after the full solve; islandCount / islandMaxSize
  broadphase        sorted permutation (broadphaseSort[1]), BroadphaseEntry[], manifolds after UpdatePairs/PackManifolds

    make -C oracle ref_full && python tests/golden/make_reference_goldens.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from phyx_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = {"stack2x50": lambda: scenes.stack(2, 50), "stack10x100": lambda: scenes.stack(10, 100),
          "falling1k": lambda: scenes.falling(1000, width=400.0, ymax=600.0)}
PACKED_WORDS = 35      # ContactJointPacked<N>: 35 words per joint (ref: Solver.h:26-45)


class Ref:
    def __init__(self, kind):
        so = os.path.join(ROOT, "oracle", "_ref", "libphyx_ref_full_%s.so" % kind)
        if not os.path.exists(so):
            raise SystemExit("%s is absent: `make -C oracle ref_full` needs /root/reference/src/microprofile/microprofile.h "
                             "(un-vendored submodule); the reference's .cpp files are unbuildable until then" % so)
        self.L = C.CDLL(so)
        self.L.reff_world_create.restype = C.c_void_p
        self.L.reff_world_create.argtypes = [C.c_float]
        for name in ("reff_world_destroy", "reff_world_pre_solve", "reff_world_solve", "reff_world_integrate_position", "reff_world_update"):
            getattr(self.L, name).restype = None
        self.L.reff_world_add_body.argtypes = [C.c_void_p] + [C.c_float] * 5 + [C.c_int]
        self.L.reff_world_pre_solve.argtypes = [C.c_void_p, C.c_float]
        self.L.reff_world_solve.argtypes = [C.c_void_p] + [C.c_int] * 4
        self.L.reff_world_integrate_position.argtypes = [C.c_void_p, C.c_float]
        self.L.reff_world_update.argtypes = [C.c_void_p, C.c_float] + [C.c_int] * 4

    def world(self, scene, gravity=-200.0):
        w = self.L.reff_world_create(gravity)
        for k in range(len(scene["px"])):
            self.L.reff_world_add_body(w, float(scene["px"][k]), float(scene["py"][k]), float(scene["angle"][k]),
                                       float(scene["sx"][k]), float(scene["sy"][k]), int(scene["static"][k]))
        return w

    def get(self, w, name, dtype, *args):
        fn = getattr(self.L, "reff_" + name)
        fn.restype = C.c_int
        n = fn(C.c_void_p(w), *args, None, 0)
        out = np.zeros(max(n, 1), dtype=dtype)
        fn(C.c_void_p(w), *args, out.ctypes.data_as(C.c_void_p), n)
        return out[:n]


def dump(kind):
    R = Ref(kind)
    for name, make in SCENES.items():
        out = {}
        scene = make()
        for k in ("px", "py", "angle", "sx", "sy", "static"):
            out["scene_" + k] = np.asarray(scene[k])
        w = R.world(scene)
        for step in (1, 2, 3):
            R.L.reff_world_pre_solve(w, 1.0 / 60.0)
            b = R.get(w, "bodies", ob.body_dtype); cp = R.get(w, "contact_points", ob.contact_point_dtype); j = R.get(w, "joints", ob.joint_dtype)
            pre = "s%d_" % step
            out[pre + "in_bodies"], out[pre + "in_contact_points"], out[pre + "in_joints"] = b.copy(), cp.copy(), j.copy()
            out[pre + "manifolds"] = R.get(w, "manifolds", ob.manifold_dtype)
            out[pre + "broadphase_sorted"] = R.get(w, "broadphase_sorted", ob.sort_entry_dtype)
            out[pre + "broadphase_entries"] = R.get(w, "broadphase_entries", ob.bp_entry_dtype)
            # grouping + refresh + per-iteration dumps: fresh worlds replayed to the same state (workers = 0 is deterministic)
            for mode, n in ((0, 1), (1, 4), (2, 8)):
                for iters in ([0, 1, 2, 5, 10, 20] if step == 1 or name == "stack2x50" else [20]):
                    w2 = R.world(scene)
                    for _ in range(step - 1):
                        R.L.reff_world_update(w2, 1.0 / 60.0, 2, 0, 20, 20)      # history in AVX2 / Single, like the main world
                    R.L.reff_world_pre_solve(w2, 1.0 / 60.0)
                    R.L.reff_world_solve(w2, mode, 0, iters, 0 if iters < 20 else 20)
                    tag = pre + "n%d_it%d_" % (n, iters)
                    out[tag + "bodies"] = R.get(w2, "bodies", ob.body_dtype)
                    out[tag + "joints"] = R.get(w2, "joints", ob.joint_dtype)
                    if iters == 0:
                        out[pre + "n%d_joint_index" % n] = R.get(w2, "joint_index", np.int32)
                        out[pre + "n%d_packed" % n] = R.get(w2, "joint_packed", np.float32, C.c_int(n)) if False else \
                            np.frombuffer(R.get(w2, "joint_packed", np.dtype(("u1", PACKED_WORDS * 4 * n)), C.c_int(n)).tobytes(), dtype=np.float32)
                    R.L.reff_world_destroy(w2)
            # Multiple island mode: partition
            w3 = R.world(scene)
            for _ in range(step - 1):
                R.L.reff_world_update(w3, 1.0 / 60.0, 2, 0, 20, 20)
            R.L.reff_world_pre_solve(w3, 1.0 / 60.0)
            R.L.reff_world_solve(w3, 2, 1, 20, 20)
            out[pre + "multiple_island_offset"] = R.get(w3, "island_offset", np.int32)
            out[pre + "multiple_island_size"] = R.get(w3, "island_size", np.int32)
            out[pre + "multiple_joint_index"] = R.get(w3, "joint_index", np.int32)
            cnt, mx = C.c_int(0), C.c_int(0)
            R.L.reff_island_stats(C.c_void_p(w3), C.byref(cnt), C.byref(mx))
            out[pre + "multiple_island_stats"] = np.array([cnt.value, mx.value], dtype=np.int32)
            out[pre + "multiple_bodies"] = R.get(w3, "bodies", ob.body_dtype)
            R.L.reff_world_destroy(w3)
            # finish the step of the main world (AVX2 / Single, 20 + 20)
            R.L.reff_world_solve(w, 2, 0, 20, 20)
            R.L.reff_world_integrate_position(w, 1.0 / 60.0)
            out[pre + "out_bodies"] = R.get(w, "bodies", ob.body_dtype)
            out[pre + "out_joints"] = R.get(w, "joints", ob.joint_dtype)
        R.L.reff_world_destroy(w)
        path = os.path.join(HERE, "reference_%s_%s.npz" % (kind, name))
        np.savez_compressed(path, **out)
        print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    for kind in ("strict", "fast"):
        dump(kind)
