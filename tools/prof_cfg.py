"""Workloads of BASELINE configs 4 and 5 for rocprofv3 (tools/gpu_prof_r3.sh): a few steady-state steps each, nothing else, so that
the per-kernel averages of a trace / PMC pass belong to that config.
usage: prof_cfg.py cfg4 | cfg2w | cfg5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phyx_amd
from phyx_amd import scenes, Configuration

which = sys.argv[1]
if which == "cfg4":                                    # 1M boxes, broadphase-heavy: whole World::Update steps
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 20, 20)
    w = phyx_amd.World(0, gravity=-200.0)
    w.add_scene(scenes.stack(10000, 100))
    for _ in range(10):
        w.Update(1.0 / 60.0, cfg)
    w.sync()
    print("cfg4", w.counts(), w.collider.stats().device_ms)
elif which == "cfg2w":                                 # the 200k-box world running: whole World::Update steps, the contact graph changing
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 20, 20)
    w = phyx_amd.World(0, gravity=-200.0)
    w.add_scene(scenes.stack(1000, 200))
    for _ in range(12):
        w.Update(1.0 / 60.0, cfg)
    w.sync()
    print("cfg2w", w.counts())
else:                                                  # 500k boxes tall stack, 50 iterations: the solver on resident inputs
    cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 50, 50)
    w = phyx_amd.World(0, gravity=-200.0)
    w.add_scene(scenes.stack(1000, 500))
    for _ in range(3):
        w.Update(1.0 / 60.0, cfg)
    w.PreSolve(1.0 / 60.0)
    arrs = [phyx_amd.DeviceArray(a, 0) for a in (w.bodies, w.contactPoints, w.contactJoints)]
    s = phyx_amd.Solver(0)
    s.bench(arrs[0], arrs[1], arrs[2], cfg, 2, 0)
    s.bench_stage(arrs[0], arrs[2], 10)
    r = s.bench(arrs[0], arrs[1], arrs[2], cfg, 0, 10)
    print("cfg5", arrs[2].count, 1e3 * r.impulse_kernel_ms / max(r.bracketed_launches, 1), "us per island launch")
