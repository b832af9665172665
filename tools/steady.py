"""Per-phase times of a running World over many steps (the topology keeps changing while a stack settles).
usage: steady.py [steps] [--no-phase-timing] ; PHX_TRACE_SCHEDULE=1 prints the schedule builder's laps to stderr."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phyx_amd
from phyx_amd import scenes, Configuration
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
w = phyx_amd.World(0, gravity=-200.0); w.set_phase_timing("--no-phase-timing" not in sys.argv); w.add_scene(scenes.stack(1000, 200))
cfg = Configuration(2, 2, 20, 20)
for step in range(steps):
    t = time.time(); w.Update(1/60, cfg); w.sync(); dt = time.time() - t
    if step % 4 == 0 or step > steps - 5:          # (the statistics are a round trip of their own: not in every step, and never in the last one —
        if step == steps - 1:                       #  the step a kernel timeline of this script is cut from)
            continue
        ss = w.solver.stats(); bs = w.collider.stats()
        print(step, "step %.2f ms" % (dt*1e3), {k: round(v, 3) for k, v in w.phase_ms().items()}, "recol", ss.recoloured, "lds groups", ss.lds_islands, "colours", ss.colour_count,
              "solve dev %.3f" % ss.device_ms, "new", bs.new_pairs, "joints", w.counts()[3], flush=True)
