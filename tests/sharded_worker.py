"""Worker of test_two_processes_share_the_gpu_and_exchange_over_gloo: one rank of an island-sharded world.
Run with RANK / WORLD_SIZE / MASTER_* set; every rank uses GPU 0 and the gloo backend (host-staged all-gather).
PHX_TEST_BACKEND=nccl (one rank only on a one-GPU box): the same through RCCL on the solver's own stream — the transport a GPU
node uses (test_exchange_through_rccl_one_rank)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import phyx_amd                                   # noqa: E402
from phyx_amd import scenes, Configuration        # noqa: E402
from phyx_amd import dist as pdist                # noqa: E402


def main():
    backend = os.environ.get("PHX_TEST_BACKEND", "gloo")
    g = pdist.init(int(os.environ["WORLD_SIZE"]), backend=backend, force=True)
    scene = scenes.stack(16, 24)
    cfg = Configuration(phyx_amd.SOLVE_SCALAR, phyx_amd.ISLAND_MULTIPLE, 12, 12)
    full = phyx_amd.World(0, gravity=-200.0)
    full.add_scene(scene)
    mine = phyx_amd.World(0, gravity=-200.0)
    mine.add_scene(scene)
    mine.set_shard(g.rank, g.world_size)
    xch = g.exchange(mine.solver, pdist.Exchange.capacity_for(len(scene["px"]), 8 * len(scene["px"])), device=0)
    for step in range(12):
        full.Update(1.0 / 60.0, cfg)
        pdist.step_sharded(mine, 1.0 / 60.0, cfg, xch)
        assert mine.bodies.tobytes() == full.bodies.tobytes(), "rank %d: bodies differ at step %d" % (g.rank, step)
        assert mine.contactJoints.tobytes() == full.contactJoints.tobytes(), "rank %d: joints differ at step %d" % (g.rank, step)
    xch.check()
    if backend == "rccl":
        # the native step: one library call per step (phx_world_step_sharded: step_begin, ncclAllGather, step_end), world-owned buffers
        native = phyx_amd.World(0, gravity=-200.0)
        native.add_scene(scene)
        native.set_comm(g.comm)
        twin = phyx_amd.World(0, gravity=-200.0)
        twin.add_scene(scene)
        for step in range(20):
            twin.Update(1.0 / 60.0, cfg)
            native.StepSharded(1.0 / 60.0, cfg)
            if step % 5 == 4:
                assert native.bodies.tobytes() == twin.bodies.tobytes(), "native sharded step: bodies differ at step %d" % step
        native.check_exchange()
        assert g.comm.async_error() == 0
    # bench.py's path: K solves queued back to back, the all-gather of every step enqueued from the step hook
    state = [phyx_amd.DeviceArray(a, 0) for a in (full.bodies, full.contactPoints, full.contactJoints)]
    mine.solver.bench_stage(state[0], state[2], 5)
    r = mine.solver.bench(state[0], state[1], state[2], cfg, 1, 5, hook=xch.hook())
    xch.check()
    assert r.impulse_iterations > 0
    groups, _ = full.solver.groups()
    assert len(groups) - 1 >= 2
    g.barrier()
    print("sharded worker ok: rank %d of %d, %d groups, %d joints" % (g.rank, g.world_size, len(groups) - 1, mine.counts()[3]))
    g.shutdown()


if __name__ == "__main__":
    main()
