"""Per-step cost of the island-sharded exchange's transport with ONE rank (what a one-GPU box can host): bench() with the all-gather
enqueued from the Python step hook through torch.distributed (backend nccl = RCCL) against the same bench with a local copy.
usage: RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 python tools/rccl_hook_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
from phyx_amd import dist as pdist
nccl_group = pdist.Group("nccl")      # (torch initialises its HIP context first: doing it after the library has used the GPU fails)
cfg = Configuration(2, phyx_amd.ISLAND_MULTIPLE, 20, 20)
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(1000, 200))
for _ in range(3): w.Update(1/60, cfg)
w.PreSolve(1/60)
d = [phyx_amd.DeviceArray(a, 0) for a in (w.bodies, w.contactPoints, w.contactJoints)]
nb, nj = d[0].count, d[2].count
for backend in ("single", "nccl"):
    g = pdist.Single() if backend == "single" else nccl_group
    s = phyx_amd.Solver(0); s.set_shard(0, 1)
    x = pdist.Exchange(g, s, pdist.Exchange.capacity_for(nb, nj) // 512 * 256, 0)
    s.bench(d[0], d[1], d[2], cfg, 0, 1, hook=x.hook())
    t = []
    for _ in range(9):
        s.bench_stage(d[0], d[2], 20); s.synchronize()
        t0 = time.perf_counter(); s.bench(d[0], d[1], d[2], cfg, 0, 20, hook=x.hook()); s.synchronize(); t.append((time.perf_counter() - t0) / 20)
    x.check()
    print("%-6s transport: %.4f ms per step (median of 9 blocks of 20 steps), segment %d bytes" % (backend, 1e3 * np.median(t), s.exchange_segment_bytes()))
