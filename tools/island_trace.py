"""Where the island kernel's time goes, per workgroup (phx_solver_set_trace): shader-clock stamps at the phase boundaries of
k_solve_islands for the cfg-2 solve.  usage: island_trace.py [columns=1000] [rows=200] [iters=20] [shards=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 200
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
shards = int(sys.argv[4]) if len(sys.argv) > 4 else 1
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(cols, rows))
cfg = Configuration(2, 2, iters, iters)
for _ in range(3): w.Update(1 / 60, cfg)
w.PreSolve(1 / 60)
b, cp, j = w.bodies, w.contactPoints, w.contactJoints
s = phyx_amd.Solver(0)
db, dc, dj = (phyx_amd.DeviceArray(a) for a in (b, cp, j))
if shards > 1: s.set_shard(0, shards)
r = s.bench(db, dc, dj, cfg, 3, 10)
plain_us = 1e3 * r.impulse_kernel_ms / max(r.bracketed_launches, 1)
s.set_trace(True, waves=False)                         # phase stamps only: the kernel keeps its speed
r = s.bench(db, dc, dj, cfg, 2, 5)
traced_us = 1e3 * r.impulse_kernel_ms / max(r.bracketed_launches, 1)
t = s.island_trace().astype(np.int64)
t = t[t[:, 0] != 0]                                   # groups of other shards never ran
t0 = t[:, 0].min()
span = (t[:, 5].max() - t0)
tick_us = 100.0                                       # wall_clock64(): constant 100 MHz
names = ["start (dispatch ramp)", "records loaded", "refreshed", "pre-stepped", "swept", "written back"]
print("island launch: %.1f us plain, %.1f us traced (HIP events); first start -> last end %.1f us; %d workgroups" % (plain_us, traced_us, span / tick_us, len(t)))
print("%-24s %10s %10s %10s %10s   (us since the first workgroup started; phase = time since the previous stamp)" % ("stamp", "min", "median", "p95", "max"))
prev = None
for k in range(6):
    at = (t[:, k] - t0) / tick_us
    line = "%-24s %10.1f %10.1f %10.1f %10.1f" % (names[k], at.min(), np.median(at), np.percentile(at, 95), at.max())
    if prev is not None:
        d = (t[:, k] - t[:, k - 1]) / tick_us
        line += "   phase: min %.1f median %.1f p95 %.1f max %.1f" % (d.min(), np.median(d), np.percentile(d, 95), d.max())
    print(line)
    prev = at
ncol = (t[:, 7] >> 32); sw = t[:, 7] & 0xffffffff
sweep_us = (t[:, 4] - t[:, 3]) / tick_us
steps = ncol * np.maximum(sw, 1)
print("colours per group: min %d median %d max %d; sweeps executed: min %d max %d" % (ncol.min(), np.median(ncol), ncol.max(), sw.min(), sw.max()))
print("us per colour step (sweeps / (colours x sweeps)): median %.3f p95 %.3f  = %.0f cycles at 2.4 GHz" % (np.median(sweep_us / steps), np.percentile(sweep_us / steps, 95), np.median(sweep_us / steps) * 2400))
ticks = t[:, 6] >> 4
wall = (t[:, 5] - t[:, 0]) / tick_us
print("s_memtime ticks per microsecond of wall clock (per workgroup): median %.1f min %.1f max %.1f" % (np.median(ticks / wall), (ticks / wall).min(), (ticks / wall).max()))
t[:, 6] &= 0xF
for x in range(8):
    m = t[:, 6] == x
    if m.any(): print("  XCC %d: %4d workgroups, start median %.1f us, end median %.1f us" % (x, m.sum(), np.median((t[m, 0] - t0) / tick_us), np.median((t[m, 5] - t0) / tick_us)))

s.set_trace(True, waves=True)                          # ... and once more with every wave's class-step cycle counts (~15 % slower)
s.bench(db, dc, dj, cfg, 1, 2)
raw = s.wave_trace()
wt = raw.astype(np.float64)
nsmall = (raw[:, :, 3] >> np.uint64(32)).astype(np.float64); nidle = (raw[:, :, 3] & np.uint64(0xffffffff)).astype(np.float64); nbig = wt[:, :, 5]
ms, mb, mi = nsmall > 0, nbig > 0, nidle > 0
print("per working colour step of a wave (shader cycles, incl. ~2 s_memtime reads):")
print("   joint update with <= 32 lanes active: median %.0f p95 %.0f (%d waves)" % (np.median(wt[:, :, 0][ms] / nsmall[ms]), np.percentile(wt[:, :, 0][ms] / nsmall[ms], 95), ms.sum()))
print("   joint update with  > 32 lanes active: median %.0f p95 %.0f (%d waves)" % (np.median(wt[:, :, 4][mb] / nbig[mb]), np.percentile(wt[:, :, 4][mb] / nbig[mb], 95), mb.sum()))
w = (nsmall + nbig) > 0
print("   barrier behind the update: median %.0f" % np.median(wt[:, :, 1][w] / (nsmall + nbig)[w]))
print("per idle colour step of a wave: median %.0f cycles" % np.median(wt[:, :, 2][mi] / nidle[mi]))
