"""Why do identical columns finish at different times?  Per group: sweep-phase time, the CU it ran on, its waves' SIMDs and their duty
(class steps in which the wave works) — against the number of groups on the CU and the busiest SIMD's summed duty.
usage: group_spread.py [columns=1000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(cols, 200))
cfg = Configuration(2, 2, 20, 20)
for _ in range(3): w.Update(1 / 60, cfg)
w.PreSolve(1 / 60)
s = phyx_amd.Solver(0)
db, dc, dj = (phyx_amd.DeviceArray(a) for a in (w.bodies, w.contactPoints, w.contactJoints))
s.set_trace(True)
s.bench(db, dc, dj, cfg, 2, 3)
t = s.island_trace().astype(np.int64)
raw = s.wave_trace()
ok = t[:, 0] != 0
t, raw = t[ok], raw[ok]
sweep_us = (t[:, 4] - t[:, 3]) / 100.0
total_us = (t[:, 5] - t[:, 0]) / 100.0
hw = raw[:, :, 6].astype(np.int64); xcc = raw[:, :, 7].astype(np.int64) & 0xF
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc[:, 0] * 8 + se[:, 0]) * 2 + sh[:, 0]) * 16 + cu[:, 0]
nsmall = (raw[:, :, 3] >> np.uint64(32)).astype(np.int64); nbig = raw[:, :, 5].astype(np.int64)
duty = nsmall + nbig                                   # class steps in which the wave worked
work_cyc = (raw[:, :, 0] + raw[:, :, 4]).astype(np.float64)
print("groups %d; sweep phase us: min %.1f median %.1f p95 %.1f max %.1f; whole kernel per group: median %.1f max %.1f" % (len(t), sweep_us.min(), np.median(sweep_us), np.percentile(sweep_us, 95), sweep_us.max(), np.median(total_us), total_us.max()))
print("working steps per wave index (median):", np.median(duty, axis=0).tolist(), " mean cycles per working step per wave index:", (work_cyc.sum(axis=0) / np.maximum(duty.sum(axis=0), 1)).round(0).tolist())
print("wave index -> SIMD histogram:")
for wv in range(raw.shape[1]): print("  wave", wv, np.bincount(simd[:, wv], minlength=4).tolist())
u, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
print("CUs used %d; groups per CU histogram %s" % (len(u), np.bincount(cnt).tolist()))
per_cu_groups = cnt[inv]
for n in np.unique(per_cu_groups):
    m = per_cu_groups == n
    print("  groups on a CU with %d groups: %4d  sweep us median %.1f p95 %.1f max %.1f" % (n, m.sum(), np.median(sweep_us[m]), np.percentile(sweep_us[m], 95), sweep_us[m].max()))
# busiest SIMD of the group's CU: summed duty of the waves that sit on it
load = np.zeros((len(u), 4))
for g in range(len(t)):
    for wv in range(raw.shape[1]): load[inv[g], simd[g, wv]] += duty[g, wv]
maxload = load.max(axis=1)[inv]; spread = (load.max(axis=1) - load.min(axis=1))[inv]
steps = duty.max()
print("class steps of a group (max duty of a wave): %d" % steps)
for lo, hi in ((0, 1.0), (1.0, 1.5), (1.5, 2.0), (2.0, 2.5), (2.5, 3.0), (3.0, 9)):
    m = (maxload / max(steps, 1) >= lo) & (maxload / max(steps, 1) < hi)
    if m.any(): print("  busiest SIMD carries %.1f-%.1f wave-duties: %4d groups, sweep us median %.1f max %.1f" % (lo, hi, m.sum(), np.median(sweep_us[m]), sweep_us[m].max()))
print("corr(sweep time, groups on CU) %.2f  corr(sweep time, busiest SIMD load) %.2f" % (np.corrcoef(sweep_us, per_cu_groups)[0, 1], np.corrcoef(sweep_us, maxload)[0, 1]))
slow = np.argsort(-sweep_us)[:6]
for g in slow:
    print("  slow group %4d: sweep %.1f us, CU key %d with %d groups, SIMDs %s duty %s; CU SIMD loads %s" % (g, sweep_us[g], key[g], per_cu_groups[g], simd[g].tolist(), duty[g].tolist(), load[inv[g]].tolist()))
fast = np.argsort(sweep_us)[:3]
for g in fast:
    print("  fast group %4d: sweep %.1f us, CU key %d with %d groups, SIMDs %s duty %s; CU SIMD loads %s" % (g, sweep_us[g], key[g], per_cu_groups[g], simd[g].tolist(), duty[g].tolist(), load[inv[g]].tolist()))
