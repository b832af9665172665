#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X (contract: see the task brief / DESIGN.md §7).

Metric (BASELINE.json): solver iterations/sec + contacts/sec on the 200k-box stack scene.  A "step" is one
Solver::SolveJoints (ref: src/Solver.cpp:17-119) over the resident solver inputs of that scene: PrepareBodies,
schedule check, PrepareJoints+RefreshJoints, PreStepJoints, `iters` impulse + displacement sweeps, FinishJoints,
FinishBodies.  `value` = joint-visits per second (contacts/sec: joints swept, skipped ones included, SURVEY.md
§8(d)) over the whole step wall time, summed over all ranks; solver iterations/sec is reported next to it.

N=1: workload = BASELINE config 2 (stack(1000,200) = 200 001 bodies, Single Sloppy islands, 20+20 iterations).
N>1: weak scaling — every rank owns its own slab of 1000 columns (= 1000 islands) of one world N*1000 columns
wide, solves it on its GPU, and the ranks meet at a 4-byte RCCL all-reduce after every step; no body or joint
data crosses ranks because islands are body-disjoint.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_IMPULSE_VISIT = 196        # algorithmic bytes per impulse joint-visit
BYTES_DISPLACEMENT_VISIT = 136   # per displacement joint-visit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--columns", type=int, default=1000, help="stack columns per GPU (1000 x 200 = config 2)")
    ap.add_argument("--rows", type=int, default=200)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--scene-steps", type=int, default=3, help="world steps run before the solver input is captured")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed Single-mode comparison run")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample budget")
    ap.add_argument("--force-dist", action="store_true", help="use the torch.distributed path even for one rank")
    ap.add_argument("--backend", default="nccl")
    args = ap.parse_args()

    from phyx_amd import dist as pdist
    group = pdist.init(args.gpus, backend=args.backend, force=args.force_dist)
    rank, world = group.rank, group.world_size
    device = group.local_rank

    import phyx_amd
    from phyx_amd import scenes, Configuration

    info = phyx_amd.device_info(device)
    # N=1: BASELINE config 2 (Single Sloppy).  N>1: config 3 (Multiple island mode, islands sharded across the GPUs).
    island_mode = phyx_amd.ISLAND_SINGLE_SLOPPY if world == 1 else phyx_amd.ISLAND_MULTIPLE
    cfg = Configuration(phyx_amd.SOLVE_AVX2, island_mode, args.iters, args.iters)

    # ---- scene: this rank's slab of the wide world, brought to a settled contact state by the product World
    first_col, ncols = pdist.shard_columns(args.columns * world, rank, world)
    scene = scenes.stack(ncols, args.rows, x_offset_columns=first_col)
    world_obj = phyx_amd.World(device, gravity=-200.0)
    world_obj.add_scene(scene)
    for _ in range(args.scene_steps):
        world_obj.Update(1.0 / 60.0, cfg)
    world_obj.PreSolve(1.0 / 60.0)
    bodies, cps, joints = world_obj.bodies, world_obj.contactPoints, world_obj.contactJoints
    nb, nj = len(bodies), len(joints)

    solver = phyx_amd.Solver(device)
    d_bodies, d_cps, d_joints = (phyx_amd.DeviceArray(a, device) for a in (bodies, cps, joints))

    hook = group.stream_hook(solver.stream_ptr())

    def run(config, warmup, steps):
        """`steps` timed solves of the resident input under `config`; returns wall seconds + HIP-event totals."""
        for _ in range(max(warmup, 1)):                               # untimed: builds the schedule, captures graphs
            solver.bench(d_bodies, d_cps, d_joints, config, 0, 1, hook=hook)
        group.barrier()
        solver.synchronize()
        t0 = time.perf_counter()
        # each step: restore the input, one full SolveJoints, then (N > 1) the per-step 4-byte RCCL all-reduce, enqueued on the
        # solver's stream so that step s+1 of every rank starts only after step s of all ranks — ordered on the device, the
        # host queues the K steps back to back inside one library call
        r = solver.bench(d_bodies, d_cps, d_joints, config, 0, steps, hook=hook)
        tot = dict(total_ms=r.total_ms, sweep_ms=r.impulse_kernel_ms, launches=r.impulse_launches, visits=r.joint_visits,
                   iterations=r.impulse_iterations)
        solver.synchronize()
        group.barrier()
        tot["elapsed"] = time.perf_counter() - t0
        tot["stats"] = solver.stats()
        return tot

    def roofline(tot, steps, kernel):
        st = tot["stats"]
        disp_visits = st.displacement_iterations * nj * steps           # upper bound: every group runs the longest count
        alg_bytes = BYTES_IMPULSE_VISIT * tot["visits"] + BYTES_DISPLACEMENT_VISIT * disp_visits
        sweep_s = tot["sweep_ms"] * 1e-3
        achieved = alg_bytes / sweep_s / 1e9 if sweep_s > 0 else 0.0
        return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "launches": tot["launches"], "avg_launch_us": 1e3 * tot["sweep_ms"] / max(tot["launches"], 1),
                "algorithmic_bytes_per_launch": alg_bytes / max(tot["launches"], 1)}

    # ---- timed region: exactly K steps of config 2, barrier + device sync on both sides (inside run())
    main_tot = run(cfg, args.warmup, args.steps)
    st = main_tot["stats"]
    elapsed_max = group.reduce_max(main_tot["elapsed"])
    visits_all = group.reduce_sum(main_tot["visits"])
    iters_all = group.reduce_sum(main_tot["iterations"])
    joints_all = group.reduce_sum(nj)
    bodies_all = group.reduce_sum(nb)

    # ---- secondary (N=1 only, untimed by the driver): strict Single island mode = the HBM colour path
    single_tot = None
    if world == 1 and not args.no_secondary:
        single_cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE, args.iters, args.iters)
        single_tot = run(single_cfg, 2, max(5, args.steps // 2))

    if rank == 0:
        ms_per_step = 1e3 * elapsed_max / max(args.steps, 1)
        lds = st.lds_islands > 0
        roof = roofline(main_tot, args.steps, "k_solve_islands (one workgroup per island, all sweeps in LDS)" if lds
                        else "k_solve_colour<impulse,displacement>")
        roof["note"] = ("achieved = ALGORITHMIC bytes (196 B per impulse joint-visit + 136 B per displacement joint-visit, SURVEY.md §8d) "
                        "over the HIP-event time of the sweep launches. " +
                        ("The island kernel keeps body state in LDS and joint constants in registers for all sweeps, so its real HBM "
                         "traffic (`traffic`, PMC) is a small fraction of the algorithmic bytes and `frac` can exceed 1: the kernel is "
                         "bound by LDS latency + workgroup barriers, not by HBM. The HBM-streaming form of the same sweeps is "
                         "`extra.single_mode.roofline`." if lds else ""))
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                roof["traffic"] = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                pass
        out = {
            "metric": "solver joint-visits/s (contacts/sec) on the 200k-box stack scene; solver iterations/s in extra",
            "value": visits_all / elapsed_max,
            "unit": "joint-visits/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("cfg2: stack(%d,%d) = %d bodies / %d joints, Single Sloppy island mode, %d+%d iterations, full SolveJoints "
                                    "per step on HBM-resident inputs" % (args.columns, args.rows, nb, nj, args.iters, args.iters)) if world == 1 else
                                   ("cfg3 (weak-scaled): Multiple island mode, islands sharded across %d GPUs as slabs of stack(%d,%d) = %d bodies / "
                                    "%d joints per GPU, %d+%d iterations, full SolveJoints per step on HBM-resident inputs, 4-byte RCCL all-reduce "
                                    "per step" % (world, args.columns, args.rows, nb, nj, args.iters, args.iters)),
                       "bodies_total": int(bodies_all), "joints_total": int(joints_all), "colours": st.colour_count,
                       "lds_islands": st.lds_islands, "graph_replay": st.graph_replay,
                       "impulse_sweeps_per_step": st.impulse_iterations, "displacement_sweeps_per_step": st.displacement_iterations,
                       "parallelism": "islands sharded by column slab, 1 rank per GPU, per-step 4-byte RCCL all-reduce" if world > 1 else "1 GPU",
                       "device": info["name"], "compute_units": info["compute_units"]},
            "extra": {"solver_iterations_per_sec": iters_all / world / elapsed_max,
                      "contacts_resolved_per_sec": joints_all * args.steps / elapsed_max,
                      "device_ms_per_step": main_tot["total_ms"] / max(args.steps, 1),
                      "sweep_ms_per_step": main_tot["sweep_ms"] / max(args.steps, 1),
                      "joint_visits_per_sec_sweeps_only": main_tot["visits"] / (main_tot["sweep_ms"] * 1e-3) if main_tot["sweep_ms"] > 0 else None},
            "roofline": roof,
        }
        if single_tot is not None:
            k = max(5, args.steps // 2)
            sst = single_tot["stats"]
            out["extra"]["single_mode"] = {
                "what": "same input, island_mode = Single (no island split): colour-by-colour sweeps out of HBM",
                "ms_per_step": 1e3 * single_tot["elapsed"] / k, "joint_visits_per_sec": single_tot["visits"] / single_tot["elapsed"],
                "colours": sst.colour_count, "impulse_sweeps_per_step": sst.impulse_iterations,
                "roofline": roofline(single_tot, k, "k_solve_colour<impulse,displacement>")}
        if world == 1 and not args.no_secondary:
            out["extra"]["other_configs"] = other_configs(phyx_amd, scenes, Configuration, device, world_obj, cfg)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(bodies, cps, joints, args.iters, args.cpu_seconds)
        print(json.dumps(out))
    group.shutdown()


def other_configs(phyx_amd, scenes, Configuration, device, cfg2_world, cfg2):
    """Untimed-by-the-driver side measurements of the other BASELINE.json configs (N=1): the whole device-resident
    World::Update at cfg 2 size, the broadphase at cfg 4 size (1M boxes) and the solver at cfg 5 size (500k boxes,
    50 iterations).  They are parity-test cases first (tests/test_*_gpu.py); the numbers here are informational."""
    res = {}
    # cfg 2, whole step (ref: World.cpp:19-37), topology still changing (new contacts every step)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); cfg2_world.FinishStep(1.0 / 60.0, cfg2); cfg2_world.PreSolve(1.0 / 60.0); cfg2_world.sync()
        t.append(time.perf_counter() - t0)
    cfg2_world.set_phase_timing(True)               # the per-phase breakdown costs a synchronisation per phase: measured separately
    cfg2_world.FinishStep(1.0 / 60.0, cfg2); cfg2_world.PreSolve(1.0 / 60.0)
    ph = cfg2_world.phase_ms()
    cfg2_world.set_phase_timing(False)
    res["cfg2_world_step"] = {"ms_per_step": 1e3 * float(np.median(t)), "phases_ms": {k: round(v, 3) for k, v in ph.items()},
                              "counts": dict(zip(("bodies", "manifolds", "contact_points", "joints"), cfg2_world.counts()))}
    # cfg 4: 1M boxes, broadphase-heavy
    w4 = phyx_amd.World(device, gravity=-200.0)
    w4.add_scene(scenes.stack(10000, 100))
    for _ in range(3):
        w4.Update(1.0 / 60.0, cfg2)
    bs = w4.collider.stats()
    w4.sync()
    t0 = time.perf_counter(); w4.Update(1.0 / 60.0, cfg2); w4.sync(); step4 = time.perf_counter() - t0
    bs = w4.collider.stats()
    # algorithmic bytes: 112 B per body for key build + radix sort + gather, 20 B per candidate test
    alg = 112.0 * w4.counts()[0] + 20.0 * bs.candidate_tests
    res["cfg4_broadphase_1M"] = {"device_ms": bs.device_ms, "candidate_tests": bs.candidate_tests, "new_pairs": bs.new_pairs,
                                 "algorithmic_GBps": alg / (bs.device_ms * 1e-3) / 1e9, "world_step_ms": 1e3 * step4,
                                 "counts": dict(zip(("bodies", "manifolds", "contact_points", "joints"), w4.counts()))}
    del w4
    # cfg 5: 500k boxes tall stack, 50 iterations, fp32 body state
    cfg5 = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE_SLOPPY, 50, 50)
    w5 = phyx_amd.World(device, gravity=-200.0)
    w5.add_scene(scenes.stack(1000, 500))
    for _ in range(3):
        w5.Update(1.0 / 60.0, cfg5)
    w5.PreSolve(1.0 / 60.0)
    arrs = [phyx_amd.DeviceArray(a, device) for a in (w5.bodies, w5.contactPoints, w5.contactJoints)]
    s5 = phyx_amd.Solver(device)
    s5.bench(arrs[0], arrs[1], arrs[2], cfg5, 2, 0)
    t0 = time.perf_counter(); r = s5.bench(arrs[0], arrs[1], arrs[2], cfg5, 0, 10); el = time.perf_counter() - t0
    st = s5.stats()
    res["cfg5_500k_tall_50it_fp32"] = {"ms_per_step": 1e3 * el / 10, "joint_visits_per_sec": r.joint_visits / el, "joints": arrs[2].count,
                                       "impulse_sweeps": st.impulse_iterations, "lds_islands": st.lds_islands, "sweep_ms_per_step": r.impulse_kernel_ms / 10}
    # the ablation of config 5: solver-side body state in fp16 (fp32 arithmetic, every store rounds to nearest even)
    def solved(solver):
        b, j = phyx_amd.DeviceArray(w5.bodies, device), phyx_amd.DeviceArray(w5.contactJoints, device)
        solver.SolveJointsDevice(b, arrs[1], j, cfg5); solver.synchronize()
        return b.to_host(), j.to_host()
    b32, j32 = solved(s5)
    s5.set_body_state_bits(16)
    s5.bench(arrs[0], arrs[1], arrs[2], cfg5, 2, 0)
    t0 = time.perf_counter(); r16 = s5.bench(arrs[0], arrs[1], arrs[2], cfg5, 0, 10); el16 = time.perf_counter() - t0
    st16 = s5.stats()
    b16, j16 = solved(s5)
    res["cfg5_500k_tall_50it_fp16_body_state"] = {
        "ms_per_step": 1e3 * el16 / 10, "joint_visits_per_sec": r16.joint_visits / el16, "impulse_sweeps": st16.impulse_iterations,
        "sweep_ms_per_step": r16.impulse_kernel_ms / 10,
        "max_abs_velocity_diff_vs_fp32": float(max(np.abs(b16["velocity"]["x"] - b32["velocity"]["x"]).max(), np.abs(b16["velocity"]["y"] - b32["velocity"]["y"]).max())),
        "mean_abs_velocity_diff_vs_fp32": float(np.abs(b16["velocity"]["y"] - b32["velocity"]["y"]).mean()),
        "max_abs_impulse_diff_vs_fp32": float(np.abs(j16["normal_acc"] - j32["normal_acc"]).max())}
    return res


def cpu_baseline(bodies, cps, joints, iters, budget_s):
    """The oracle's impulse loop (restated ref: Solver.cpp:760-914) timed on this host over the same solver input,
    Single-Sloppy style: persistent threads, 512-joint batches (ref: Solver.cpp:138-139).  The thread count is the
    best of a short probe over {1, 4, 8, 16, 32, 64, all} — the racy sweep stops scaling long before 256 threads.
    Reported beside the GPU number, not a target."""
    from oracle import binding as ob
    ncpu = os.cpu_count() or 1
    b = bodies.view(ob.body_dtype); cp = cps.view(ob.contact_point_dtype); j = joints.view(ob.joint_dtype)
    t_begin = time.perf_counter()
    probe = {}
    for t in sorted({1, 4, 8, 16, 32, 64, ncpu}):
        if t > ncpu:
            continue
        sec, v = ob.time_impulse_loop(b, cp, j, iters, t)
        probe[t] = v / sec
    best = max(probe, key=probe.get)
    used = time.perf_counter() - t_begin
    one = len(j) * iters / probe[best]
    reps = max(3, min(400, int(max(budget_s - used, 1.0) / max(one, 1e-6))))
    tt = vv = 0.0
    for _ in range(reps):
        sec, v = ob.time_impulse_loop(b, cp, j, iters, best)
        tt += sec; vv += v
    return {"value": vv / tt, "unit": "joint-visits/s", "cores": best, "kind": "port",
            "sample": "%d x (%d impulse sweeps over the same %d-joint solver input), impulse loop only, %d threads in 512-joint "
                      "batches (best of probe %s; host has %d cores)" % (reps, iters, len(j), best,
                                                                         {k: round(v / 1e6) for k, v in probe.items()}, ncpu),
            "single_thread_value": probe.get(1)}


if __name__ == "__main__":
    main()
