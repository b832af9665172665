"""Which SIMD do the island kernel's waves sit on?  (HW_ID of every wave of the traced kernel: tools/island_trace.py's companion)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
cfg = Configuration(2, 2, 20, 20)
for _ in range(3): w.Update(1 / 60, cfg)
w.PreSolve(1 / 60)
s = phyx_amd.Solver(0)
db, dc, dj = (phyx_amd.DeviceArray(a) for a in (w.bodies, w.contactPoints, w.contactJoints))
s.set_trace(True)
s.bench(db, dc, dj, cfg, 1, 1)
raw = s.wave_trace()
hw = raw[:, :, 6].astype(np.int64); xcc = raw[:, :, 7].astype(np.int64)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; wave = hw & 15
print("groups", raw.shape[0], "waves per group", raw.shape[1])
print("SIMD of wave w (first 12 groups):"); print(simd[:12])
same = (simd == simd[:, :1]).all(axis=1).mean()
print("fraction of groups whose waves all sit on one SIMD: %.2f" % same)
print("wave index -> SIMD histogram:")
for wv in range(raw.shape[1]): print("  wave", wv, np.bincount(simd[:, wv], minlength=4))
key = ((xcc[:, 0] * 8 + se[:, 0]) * 2 + sh[:, 0]) * 16 + cu[:, 0]
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc, se, sh, cu):", len(u), "groups per CU: min %d max %d" % (cnt.min(), cnt.max()))
k0 = u[0]; g = np.nonzero(key == k0)[0]
print("groups on one CU:", g.tolist()); print(" their waves' SIMDs:"); print(simd[g])
