import os, sys, time
sys.path.insert(0, "/root/repo")
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.set_phase_timing(True); w.add_scene(scenes.stack(10000, 100))
cfg = Configuration(2, 2, 20, 20)
for step in range(8):
    t = time.time(); w.Update(1/60, cfg); w.sync(); dt = time.time() - t
    ss = w.solver.stats(); bs = w.collider.stats()
    print(step, "step %.2f ms" % (dt*1e3), {k: round(v, 3) for k, v in w.phase_ms().items()}, "lds groups", ss.lds_islands, "colours", ss.colour_count, "solve dev %.3f" % ss.device_ms, "bp dev %.3f" % bs.device_ms, "joints", w.counts()[3], flush=True)
