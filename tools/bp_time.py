import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 200)
w = phyx_amd.World(0, gravity=-200.0); w.set_phase_timing(True); w.add_scene(scenes.stack(nx, ny))
cfg = Configuration(2, 2, 20, 20)
for step in range(5):
    t = time.time(); w.Update(1/60, cfg); w.sync(); dt = time.time() - t
    bs = w.collider.stats(); ss = w.solver.stats()
    print(step, "step %.1f ms" % (dt*1e3), {k: round(v, 2) for k, v in w.phase_ms().items()}, "bp dev ms %.3f" % bs.device_ms, "tests", bs.candidate_tests, "new", bs.new_pairs, "set", bs.set_size, "solve dev ms %.3f" % ss.device_ms, w.counts())
