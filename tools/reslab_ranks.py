"""Ownership-sharded world across real processes: N ranks (one process each; they may share one GPU: --backend gloo) step
dist.SlabWorld over a pile that spreads sideways, so that bodies reach their slabs' boundaries and the ranks re-slab (all-gather of
the worlds' states, new cuts, phx_world_set_state) — through the same collectives a multi-GPU run uses.  Rank 0 prints one JSON
line: steps, re-slabs, the dynamic bodies every rank ended with, whether every guard holds.
usage: python tools/reslab_ranks.py --ranks 3 [--backend gloo|rccl|nccl] [--steps 40] [--every K]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--steps", type=int, default=160)
    ap.add_argument("--every", type=int, default=0, help="re-slab every K steps whatever the guards say")
    ap.add_argument("--per-cluster", type=int, default=60)
    ap.add_argument("--python", action="store_true", help="re-slab through the numpy statement of the hand-over (dist.SlabWorld.reslab_python) instead of phx_world_reslab")
    args = ap.parse_args()
    from phyx_amd import dist as pdist
    if "WORLD_SIZE" not in os.environ and args.ranks > 1:
        sys.exit(pdist.self_launch(args.ranks, timeout_s=300.0))
    import phyx_amd
    from phyx_amd import scenes
    group = pdist.init(args.ranks, backend=args.backend, force=True)
    timer = pdist.watchdog(240.0, lambda: sys.stderr.write("reslab_ranks: rank %d timed out\n" % group.rank))
    device = group.local_rank if args.backend in ("rccl", "nccl") else 0
    scene = scenes.piles(2 * args.ranks, args.per_cluster, pitch=64.0, ymax=220.0)
    cfg = phyx_amd.Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE, 10, 6)
    sw = pdist.SlabWorld(group, scene, device=device, gravity=-200.0, reslab_every=args.every)
    sw.python_reslab = args.python
    for _ in range(args.steps):
        sw.step(1.0 / 60.0, cfg)
    bodies = sw.world.bodies
    dyn = int(np.count_nonzero(bodies["inv_mass"] > 0))
    total = int(group.reduce_sum(dyn))
    worst = int(group.step_barrier_value(0 if sw.inside() else 1))
    full = sw.gather_bodies()
    import hashlib
    if group.rank == 0:
        print(json.dumps({"digest": hashlib.sha256(full.tobytes()).hexdigest(), "ranks": group.world_size, "steps": sw.steps, "reslabs": sw.reslabs, "dynamic_bodies_total": total,
                          "dynamic_bodies_scene": int(np.count_nonzero(~np.asarray(scene["static"], dtype=bool))),
                          "dynamic_bodies_rank0": dyn, "every_guard_holds": worst == 0,
                          "finite": bool(np.isfinite(full["pos"]["x"]).all() and np.isfinite(full["pos"]["y"]).all()),
                          "backend": getattr(group, "backend", args.backend)}))
    timer.cancel()
    group.shutdown()


if __name__ == "__main__":
    main()
