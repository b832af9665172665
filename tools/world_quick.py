"""cfg-2 world step time (median of N steps after 4; N = argv[1], default 9) and broadphase device time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
cfg = Configuration(2, 2, 20, 20)
for _ in range(4): w.Update(1/60, cfg)
w.sync()
t = []
import ctypes as C
from phyx_amd import _lib
L = _lib.load(); ns0, c0 = C.c_longlong(0), C.c_longlong(0); L.phx_debug_wait_clock(C.byref(ns0), C.byref(c0))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 9):
    t0 = time.perf_counter(); w.Update(1/60, cfg); w.sync(); t.append(time.perf_counter() - t0)
print("world step median %.3f ms (min %.3f, mean %.3f over %d), broadphase device %.3f ms" % (1e3 * float(np.median(t)), 1e3 * min(t), 1e3 * float(np.mean(t)), len(t), w.collider.stats().device_ms))
lite, full = w.build_counts()
print("schedule rebuilds: %d took components and bins from the manifolds (side stream), %d from the joints" % (lite, full))
if "-v" in sys.argv: print("per step ms:", " ".join("%.3f" % (1e3 * x) for x in t))
ns1, c1 = C.c_longlong(0), C.c_longlong(0); L.phx_debug_wait_clock(C.byref(ns1), C.byref(c1))
if c1.value > c0.value: print("host waits per step: %.1f, %.3f ms of the step spent waiting (PHX_WAIT_CLOCK)" % ((c1.value - c0.value) / len(t), 1e-6 * (ns1.value - ns0.value) / len(t)))
