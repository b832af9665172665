#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X (contract: see the task brief / DESIGN.md §7).

Metric (BASELINE.json): solver iterations/sec + contacts/sec on the 200k-box stack scene.  A "step" is one
Solver::SolveJoints (ref: src/Solver.cpp:17-119) over the solver inputs of that scene, resident in HBM in the layout
the World keeps them in (bodies as structure of arrays, csrc/body_view.h; joints and contact points as the
reference's records): PrepareJoints+RefreshJoints, PreStepJoints, `iters` impulse + displacement sweeps, FinishJoints,
FinishBodies, on the CACHED schedule (built in the warm-up; every timed solve checks it against the arrays inside the
island kernel and would rebuild on a difference).  `value` = joint-visits per second (contacts/sec: joints swept,
skipped ones included) over the whole step wall time, summed over all ranks; solver iterations/sec is
reported next to it.  The same solve with the schedule rebuilt in every step (what the reference's SolveJoints does,
ref: Solver.cpp:77, 135) and in strict Single island mode are printed as `live_topology` and `single_mode`, siblings of
`value`.

N=1: workload = BASELINE config 2 (stack(1000,200) = 200 001 bodies, Single Sloppy islands, 20+20 iterations).
N>1: workload = BASELINE config 3 — the SAME 200 001-body world on every rank, Multiple island mode, the schedule's
groups (islands binned per workgroup) sharded g % N over the ranks; after its solve every rank packs its results,
ONE all-gather per step (RCCL over xGMI, on the solver's stream) carries them to every other rank, and every rank
scatters them into its replica (csrc/exchange.h) — so after each step all N replicas hold the whole solved world.
Strong scaling: total work is fixed.

The K-step timed block (barrier + device sync on both sides) is repeated `--repeats` times and the MEDIAN block is
reported (20 steps of 0.15 ms are 3 ms: one scheduler hiccup would move a single block by several per cent).

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
SHADER_CLOCK_HZ = 2.4e9          # same guide: max clock; cycle figures below are seconds x this
BYTES_IMPULSE_VISIT = 196        # algorithmic bytes per impulse joint-visit
BYTES_DISPLACEMENT_VISIT = 136   # per displacement joint-visit
# Latency floor of one class step of the island kernel: a lane sweeps the two joints of a unit on one LDS
# read and one LDS write of the two bodies.  LDS read (issue -> use ~64 cycles), per joint the dependent fp32 chain of one
# impulse update (normal: 6 subtractions, multiply, clamp, 2-op body update; friction the same: ~26 dependent ops x ~4
# cycles), the LDS write-back (~13 cycles issue for 16 bytes) and one workgroup barrier (~128 cycles measured for an empty
# step, tools/probe/colour_step.hip)
# round 3's floor priced the class step as the unit's dependent chain: LDS read 64 + 2 x 26 dependent fp32 ops x 4 + LDS write 13 +
# barrier 128.  Round 4 measured what one wave can do (profiles/r04_unit_issue_probe.txt, profiles/r04_sq_islands.json): a wave64
# issues ONE instruction per ~4 cycles whether or not it depends on the last one, and the working wave of a class step issues
# ~222 VALU instructions (SQ_INSTS_VALU per group and class step) — so the floor of a class step is that many issue slots.
# Round 5 (profiles/r06_sq_islands.json): 164 VALU instructions per group and class step, all four waves and the first sweep's
# displacement half included; the working wave of an impulse-only step issues ~112 of them.
COLOUR_STEP_CHAIN_FLOOR_CYCLES = 64 + 2 * 26 * 4 + 13 + 128
COLOUR_STEP_VALU_INSTRUCTIONS = 105          # round 5: a class step of the impulse half in fused arithmetic (island_kernel.h half_step)
# ... and what bounds the sweeps with four groups on every CU (round 5, tools/probe/twin_units.diff.txt): not the chain of steps but the
# SIMDs' instruction issue — a class costs one pass (~COLOUR_STEP_VALU_INSTRUCTIONS wave64 instructions at one per ~4.5 cycles) of every
# WAVE it has a lane in, whatever the number of lanes at work
ISSUE_CYCLES_PER_INSTRUCTION = 4.5
SIMDS = 1024
COLOUR_STEP_FLOOR_CYCLES = 64 + COLOUR_STEP_VALU_INSTRUCTIONS * 4 + 13 + 128


def pmc_file(name):
    """kernels -> HBM bytes per launch (corrected) of a committed PMC summary profiles/<name>.json, or {}."""
    path = os.path.join(ROOT, "profiles", name + ".json")
    try:
        d = json.load(open(path))
        if d.get("kernel_source_sha256") != kernel_source_sha256():      # taken of another version of the island kernel: not quoted
            return {}
        return {"file": "profiles/%s.json" % name, "kernels": {k: v["hbm_bytes_per_launch_corrected"] for k, v in d.get("kernels", {}).items()}}
    except Exception:
        return {}


def kernel_source_sha256():
    """Fingerprint of the island kernel's source (what a committed PMC pass was taken of): island_kernel.h + solver_kernels.h."""
    import hashlib
    h = hashlib.sha256()
    for f in ("island_kernel.h", "solver_kernels.h"):
        with open(os.path.join(ROOT, "phyx_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def steady_step_file(name):
    """profiles/<name>.json (tools/steady_step_summary.py): per kernel of ONE profiled steady World::Update its launches, time and HBM bytes."""
    path = os.path.join(ROOT, "profiles", name + ".json")
    try:
        d = json.load(open(path))
        d["file"] = "profiles/%s.json" % name
        return d
    except Exception:
        return {}


def pmc_traffic():
    """HBM bytes per launch of the solve kernels from the committed PMC passes (tools/gpu_prof.sh -> tools/pmc_summary.py):
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same script, read side corrected x2 as MI355X_MICROARCH.md §HBM says.
    A pass is only as good as the kernel it profiled: the file records the source fingerprint and the commit it was taken at, and a
    file taken of another island_kernel.h / solver_kernels.h is REFUSED (`stale`) rather than quoted."""
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json")
        if os.path.exists(path):
            try:
                d = json.load(open(path))
                if d.get("kernel_source_sha256") != kernel_source_sha256():
                    return {"stale": "profiles/%s_pmc_traffic.json was taken of another version of the island kernel (commit %s): not quoted"
                                     % (tag, d.get("commit", "unrecorded"))}
                out = {"file": "profiles/%s_pmc_traffic.json" % tag, "commit": d.get("commit")}
                for name, k in d.get("kernels", {}).items():
                    if "k_solve_islands<256" in name:
                        out["k_solve_islands"] = k["hbm_bytes_per_launch_corrected"]
                if d.get("hbm_colour_bytes_per_launch"):          # mean over the Single-mode measurement's class launches
                    out["k_solve_colour"] = d["hbm_colour_bytes_per_launch"]
                    if "k_solve_dataflow" in name:
                        out["k_solve_dataflow"] = k["hbm_bytes_per_launch_corrected"]
                return out
            except Exception:
                pass
    return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=11, help="timed K-step blocks; the median block is reported")
    ap.add_argument("--columns", type=int, default=1000, help="stack columns (1000 x 200 = configs 2 and 3)")
    ap.add_argument("--rows", type=int, default=200)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--scene-steps", type=int, default=3, help="world steps run before the solver input is captured")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--secondary", action="store_true", help="also run the long side measurements (Single mode, the island kernel's phases, one rank of N, "
                    "the settled world, cfg 4, cfg 5): minutes; the builder's full line is kept under profiles/")
    ap.add_argument("--no-secondary", action="store_true", help="skip every side measurement, also the two a default run keeps (live topology, cfg 2 World::Update)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample budget")
    ap.add_argument("--force-dist", action="store_true", help="use the torch.distributed path even for one rank")
    ap.add_argument("--mode", default="slab", choices=("slab", "replica"),
                    help="N > 1: slab = ownership sharding, every rank simulates the islands of its own x-slab and the per-step collective is a "
                         "4-byte all-reduce (the north star's wording); replica = every rank holds the whole world, solves its share of the "
                         "islands and all-gathers everybody's results every step")
    ap.add_argument("--backend", default="rccl", help="rccl: the library's own RCCL transport (csrc/comm.hip), torch only for the rendezvous; "
                    "nccl: torch.distributed's ProcessGroupNCCL from a step hook; gloo: host-staged (ranks sharing one GPU)")
    args = ap.parse_args()

    from phyx_amd import dist as pdist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: one process per GPU, spawned here (python -m torch.distributed.run works too)
        sys.exit(pdist.self_launch(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return run_bench(args, pdist)
    # N > 1: nobody has ever watched this path on more than one GPU, so it must not be able to hang its launcher.  Every blocking
    # wait on a peer is bounded (PHX_COMM_TIMEOUT_S: the rendezvous, ncclCommInitRank, the library's host-side waits), a watchdog
    # bounds the whole run (PHX_BENCH_TIMEOUT_S, default 900 s), and whatever goes wrong ends as ONE JSON line with an "error"
    # field from rank 0 and a non-zero exit status.
    rank = int(os.environ.get("RANK", "0"))

    def error_line(msg):
        if rank == 0:
            print(json.dumps({"metric": "solver joint-visits/s (contacts/sec) on the 200k-box stack scene; solver iterations/s in extra",
                              "value": None, "unit": "joint-visits/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                              "data": "synthetic", "config": {"workload": "cfg3 (not completed)", "mode": args.mode, "transport": args.backend},
                              "error": msg}), flush=True)
        sys.stderr.write("bench.py rank %d: %s\n" % (rank, msg))
    try:
        limit = float(os.environ.get("PHX_BENCH_TIMEOUT_S", "0")) or 900.0
    except ValueError:
        limit = 900.0
    dog = pdist.watchdog(limit, lambda: error_line("rank %d made no progress for %.0f s (PHX_BENCH_TIMEOUT_S): a peer is missing or a collective hangs" % (rank, limit)))
    try:
        run_bench(args, pdist)
    except BaseException as e:                     # (SystemExit from dist.init included)
        dog.cancel()
        if isinstance(e, SystemExit) and not e.code:
            raise
        error_line("%s: %s" % (type(e).__name__, e))
        sys.stdout.flush()
        os._exit(1)                                # (not sys.exit: torch's / RCCL's teardown may wait for the peers that caused this)
    dog.cancel()


def run_bench(args, pdist):
    group = pdist.init(args.gpus, backend=args.backend, force=args.force_dist)
    rank, world = group.rank, group.world_size
    device = group.local_rank if getattr(group, "backend", "rccl") in ("nccl", "rccl") else 0

    import phyx_amd
    from phyx_amd import scenes, Configuration

    info = phyx_amd.device_info(device)
    # side measurements: 0 = none, 1 = the two that say what a running world pays (live topology, cfg 2 World::Update; default), 2 = all
    side = 0 if args.no_secondary else (2 if args.secondary else 1)
    # N=1: BASELINE config 2 (Single Sloppy).  N>1: config 3 (Multiple island mode, islands sharded across the GPUs).
    island_mode = phyx_amd.ISLAND_SINGLE_SLOPPY if world == 1 else phyx_amd.ISLAND_MULTIPLE
    cfg = Configuration(phyx_amd.SOLVE_AVX2, island_mode, args.iters, args.iters)

    # ---- scene, brought to a settled contact state by the product World.  N = 1 and replica mode: the whole 200k-box world on
    #      every rank.  Slab mode (N > 1): this rank's x-slab of it — its share of the columns (= islands) plus the static ground
    mode = "single" if world == 1 else args.mode
    full_scene = scenes.stack(args.columns, args.rows)
    scene = pdist.slab_partition(full_scene, world)[rank][0] if mode == "slab" else full_scene
    world_obj = phyx_amd.World(device, gravity=-200.0)
    world_obj.add_scene(scene)
    for _ in range(args.scene_steps):
        world_obj.Update(1.0 / 60.0, cfg)
    world_obj.PreSolve(1.0 / 60.0)
    bodies, cps, joints = world_obj.bodies, world_obj.contactPoints, world_obj.contactJoints
    nb, nj = len(bodies), len(joints)
    nb_total = int(group.reduce_sum(nb - 1)) + 1 if mode == "slab" else nb          # (every slab carries the ground)
    nj_total = int(group.reduce_sum(nj)) if mode == "slab" else nj

    solver = phyx_amd.Solver(device)
    d_bodies, d_cps, d_joints = (phyx_amd.DeviceArray(a, device) for a in (bodies, cps, joints))
    xch = None
    if mode == "replica":
        solver.set_shard(rank, world)
        xch = group.exchange(solver, pdist.Exchange.capacity_for(nb, nj))
    # per-step collective: replica = the all-gather of everybody's results (run by the library between pack and unpack, or from
    # this hook with the torch transports); slab = a 4-byte all-reduce queued on the solver's stream behind every step
    hook = xch.hook() if xch else (group.stream_hook(solver.stream_ptr()) if mode == "slab" else None)

    def run(config, warmup, steps, repeats, slv=solver, hk=hook):
        """`repeats` timed blocks of `steps` solves of the resident input under `config`; returns the median block."""
        for _ in range(max(warmup, 1)):                               # untimed: builds the schedule
            slv.bench(d_bodies, d_cps, d_joints, config, 0, 1, hook=hk)
        # what the timed solves must compute: the checksum of the warm-up solve's results (velocities, displacing velocities, impulses)
        want_sum = slv.bench_checksum()
        sums = []
        blocks = []
        for _ in range(max(repeats, 1)):
            slv.bench_stage(d_bodies, d_joints, steps)                    # K copies of the input, resident in HBM before the clock starts
            group.barrier()
            slv.synchronize()
            t0 = time.perf_counter()
            # each step: one full SolveJoints of this rank's groups on that step's copy of the input, then (N > 1) pack ->
            # all-gather -> unpack, all queued on the solver's stream; the host queues the K steps back to back inside one library call
            r = slv.bench(d_bodies, d_cps, d_joints, config, 0, steps, hook=hk)
            slv.synchronize()
            group.barrier()
            el = time.perf_counter() - t0
            sums.append(slv.bench_checksum())                             # (outside the clock: the last timed step's results)
            blocks.append(dict(elapsed=el, total_ms=r.total_ms, sweep_ms=r.impulse_kernel_ms, launches=r.impulse_launches, bracketed=r.bracketed_launches,
                               visits=r.joint_visits, iterations=r.impulse_iterations))
        # the block every rank reports must be the same one: rank by the max-over-ranks time
        times = [group.reduce_max(b["elapsed"]) for b in blocks]
        mid = int(np.argsort(times)[len(times) // 2])
        tot = dict(blocks[mid])
        tot["elapsed_max"] = times[mid]
        tot["all_blocks_ms_per_step"] = [1e3 * t / max(steps, 1) for t in times]
        tot["stats"] = slv.stats()
        tot["result_check"] = {"what": "64-bit checksum of the last timed step's velocities, displacing velocities and accumulated impulses against the "
                                       "warm-up solve's, every timed block, outside the clock (same input, same schedule: bit-equal)",
                               "checksum": "%016x" % want_sum, "blocks_checked": len(sums), "equal": all(x == want_sum for x in sums)}
        if not tot["result_check"]["equal"]:
            raise RuntimeError("bench: a timed block's results differ from the warm-up solve's (checksums %s, expected %016x)" % (["%016x" % x for x in sums], want_sum))
        return tot

    def algorithmic_bytes(tot, steps):
        st = tot["stats"]
        disp_visits = st.displacement_iterations * nj * steps           # upper bound: every group runs the longest count
        return BYTES_IMPULSE_VISIT * tot["visits"] + BYTES_DISPLACEMENT_VISIT * disp_visits

    # ---- timed region: K steps, barrier + device sync on both sides (inside run()), median of `repeats` blocks
    main_tot = run(cfg, args.warmup, args.steps, args.repeats)
    st = main_tot["stats"]
    elapsed_max = main_tot["elapsed_max"]
    visits_all = group.reduce_sum(main_tot["visits"])
    iters_max = group.reduce_max(main_tot["iterations"])
    exchange_status = 0
    if xch:
        exchange_status = int(group.reduce_max(solver.exchange_status()))

    traffic = pmc_traffic()
    extra = {}
    lds = st.lds_islands > 0

    # ---- the island kernel's phases, measured live: a 0-iteration solve is set-up + PreStep + write-back only
    phases = None
    rank_of_8 = None
    if world == 1 and lds and side >= 1:
        zero = run(Configuration(phyx_amd.SOLVE_AVX2, island_mode, 0, 0), 2, 10, 3)
        phases = {"setup_prestep_writeback_us": 1e3 * zero["sweep_ms"] / max(zero["bracketed"], 1)}
        # one rank of eight on this GPU (an eighth of the groups, <= 1 per CU, an eighth of the instructions): if the launch hardly gets
        # shorter, the chain of class steps bounds it, not the SIMDs' issue and not HBM
        try:
            rank_of_8 = one_rank_of_n(phyx_amd, Configuration, group, solver, d_bodies, d_cps, d_joints, args, nb, nj, run, ns=(8,))["n=8"]
        except Exception as e:      # (a diagnostic: never the reason a bench line is missing)
            rank_of_8 = {"error": str(e)}
        # ... and the cost of a class step as a SLOPE: the same solve with half the impulse sweeps.  (The 0-iteration launch is not
        # "the full launch minus its sweeps": its workgroups all reach the write-back together, and under in-kernel verification
        # they wait there for the last arrival, which a full launch hides behind its sweeps.)
        half_it = max(cfg.contactIterationsCount // 2, 1)
        half = run(Configuration(phyx_amd.SOLVE_AVX2, island_mode, half_it, cfg.penetrationIterationsCount), 2, 10, 3)
        if half["stats"].impulse_iterations == half_it and st.impulse_iterations > half_it:
            phases["half_sweeps"] = half_it
            phases["half_sweeps_launch_us"] = 1e3 * half["sweep_ms"] / max(half["bracketed"], 1)

    # ---- secondary (N=1 only, untimed by the driver): strict Single island mode = the general-case (big island) path
    single_tot = live_tot = unbracketed = None
    if world == 1 and side >= 2:
        os.environ["PHX_BENCH_BRACKET_STRIDE"] = "0"               # the same block with no HIP event inside the timed region at all
        unbracketed = run(cfg, 1, args.steps, 3)
        del os.environ["PHX_BENCH_BRACKET_STRIDE"]
        single_cfg = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_SINGLE, args.iters, args.iters)
        single_tot = run(single_cfg, 2, max(5, args.steps // 2), 3)
    if world == 1 and side >= 1:
        # live topology: the schedule is rebuilt inside the timed region on every solve, like the reference rebuilds
        # PrepareIndices / GatherIslands on every call (ref: Solver.cpp:77, 135)
        solver.set_schedule_reuse(False)
        live_tot = run(cfg, 2, args.steps, 3)
        solver.set_schedule_reuse(True)

    # ---- N > 1 side measurement: the OTHER sharding mode on the same 200k-box world (slab <-> replica), same strong-scaling accounting
    other_mode = None
    if world > 1 and side >= 2:
        other_mode = measure_other_mode(phyx_amd, scenes, Configuration, pdist, group, device, args, cfg, "replica" if mode == "slab" else "slab", full_scene)

    # ---- N > 1 side measurement (round-1 headline, kept for comparison): weak scaling — every rank solves its OWN slab of 1000
    #      columns of one N*1000-column world, no data crosses ranks, the ranks meet at a 4-byte all-reduce per step
    weak = None
    if world > 1 and side >= 2:
        first_col, ncols = pdist.shard_columns(args.columns * world, rank, world)
        wslab = phyx_amd.World(device, gravity=-200.0)
        wslab.add_scene(scenes.stack(ncols, args.rows, x_offset_columns=first_col))
        for _ in range(args.scene_steps):
            wslab.Update(1.0 / 60.0, cfg)
        wslab.PreSolve(1.0 / 60.0)
        arrs = [phyx_amd.DeviceArray(a, device) for a in (wslab.bodies, wslab.contactPoints, wslab.contactJoints)]
        wsolver = phyx_amd.Solver(device)
        whook = group.stream_hook(wsolver.stream_ptr())
        for _ in range(2):
            wsolver.bench(arrs[0], arrs[1], arrs[2], cfg, 0, 1, hook=whook)
        group.barrier(); wsolver.synchronize()
        t0 = time.perf_counter()
        r = wsolver.bench(arrs[0], arrs[1], arrs[2], cfg, 0, args.steps, hook=whook)
        wsolver.synchronize(); group.barrier()
        wel = group.reduce_max(time.perf_counter() - t0)
        wvis = group.reduce_sum(r.joint_visits)
        weak = {"what": "every rank solves its own 1000-column slab (%d joints here), per-step 4-byte all-reduce, no data exchange" % arrs[2].count,
                "ms_per_step": 1e3 * wel / max(args.steps, 1), "joint_visits_per_sec": wvis / wel, "scaling": "weak"}

    if rank == 0:
        ms_per_step = 1e3 * elapsed_max / max(args.steps, 1)
        launches = max(main_tot["launches"], 1)
        launch_us = 1e3 * main_tot["sweep_ms"] / max(main_tot["bracketed"], 1)      # HIP events around the launches of every 4th timed step
        alg = algorithmic_bytes(main_tot, args.steps) / launches
        kname = "k_solve_islands" if lds else "k_solve_colour"
        tbytes = traffic.get(kname)
        tsource = traffic.get("file") or traffic.get("stale")
        if traffic.get("file"):
            tsource += " (a committed rocprofv3 --pmc pass of this script at commit %s; NOT measured in this run)" % traffic.get("commit")
        # the PMC passes profiled the default workload at N = 1: a launch of rank 0 at N > 1 covers only its own groups (its share
        # of the joint visits), and any other scene size was not profiled at all
        if (args.columns, args.rows) != (1000, 200):
            tbytes, tsource = None, "not profiled for this scene size"
        elif tbytes and world > 1:
            tbytes *= main_tot["visits"] / max(visits_all, 1)
            tsource = "%s, scaled by rank 0's share of the joint visits (1 of %d ranks)" % (tsource, world)
        roof = {"bound": "hbm",
                "kernel": "k_solve_islands<256,768> (one workgroup per island group, one lane per unit of two joints, all sweeps in LDS)" if lds else "k_solve_colour<impulse,displacement>",
                "achieved": (tbytes / (launch_us * 1e-6) / 1e9) if tbytes else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (tbytes / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if tbytes else None,
                "traffic": tbytes, "traffic_source": tsource,
                "launches": main_tot["launches"], "launches_bracketed_by_hip_events": main_tot["bracketed"], "avg_launch_us": launch_us,
                "algorithmic_bytes_per_launch": alg, "algorithmic_GBps": alg / (launch_us * 1e-6) / 1e9,
                "note": ("achieved / frac = HBM bytes the kernel really moves per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, `traffic`) over its "
                         "live HIP-event launch time.  algorithmic_GBps = SURVEY.md §8(d) bytes (196 B per impulse joint-visit, 136 B per "
                         "displacement joint-visit, no reuse assumed) over the same time; it exceeds the HBM peak because the kernel keeps "
                         "body state in LDS and joint constants in registers across all sweeps — it is a statement about avoided traffic, "
                         "not a bandwidth claim.  The kernel is latency-bound: see latency_model.")}
        if lds:
            # colour steps on the critical path: the slowest group runs (iterations + PreStep) x its colour count
            ncol_max = int(world_groups_max_colours(solver))
            steps_crit = ncol_max * (st.impulse_iterations + 1)
            model = {"colour_steps_on_critical_path": steps_crit, "colours_of_the_slowest_group": ncol_max,
                     "floor_cycles_per_colour_step": COLOUR_STEP_FLOOR_CYCLES,
                     "floor_what": "class step of the one working wave: LDS read 64 + %d VALU instructions x 4 cycles of issue (measured: one wave64 issues an "
                                   "instruction per ~4 cycles dependent or not, profiles/r04_unit_issue_probe.txt; instructions of the hot form of a class step, "
                                   "island_kernel.h half_step; SQ counters of the whole launch: profiles/r06_sq_islands.json) + LDS write 13 + barrier 128 cycles" % COLOUR_STEP_VALU_INSTRUCTIONS,
                     "dependent_chain_floor_cycles_round3": COLOUR_STEP_CHAIN_FLOOR_CYCLES}
            try:
                # wave passes per sweep over all groups, from the host-side statement of the same schedule (phx_schedule_groups: classes
                # and lanes are a pure function of the joints; the device builder is tested equal to it)
                hs = phyx_amd.schedule_groups(joints["body1"], joints["body2"], ((bodies["inv_mass"] == 0) & (bodies["inv_inertia"] == 0)).astype(np.uint8),
                                              joints["contact_point_index"])
                cls_of_slot = np.searchsorted(hs["colour_offsets"], hs["unit_leader_slot"], side="right") - 1
                grp_of_slot = np.searchsorted(hs["group_offsets"], hs["unit_leader_slot"], side="right") - 1
                waves = np.unique(np.stack([grp_of_slot, cls_of_slot, hs["unit_lane"] // 64], axis=1), axis=0)
                passes = int(len(waves))
                floor_us = passes * COLOUR_STEP_VALU_INSTRUCTIONS * ISSUE_CYCLES_PER_INSTRUCTION / SIMDS / SHADER_CLOCK_HZ * 1e6
                model["issue_model"] = {"wave_passes_per_sweep_all_groups": passes, "groups": int(hs["lds_groups"]),
                                        "wave_passes_per_sweep_of_a_group": passes / max(int(hs["lds_groups"]), 1),
                                        "valu_instructions_per_pass": COLOUR_STEP_VALU_INSTRUCTIONS, "cycles_per_issue": ISSUE_CYCLES_PER_INSTRUCTION, "simds": SIMDS,
                                        "issue_floor_us_per_sweep": floor_us, "issue_floor_us_all_sweeps": floor_us * st.impulse_iterations,
                                        "what": "every class is one pass of each wave it has a lane in: passes x instructions x cycles per issue, spread over the "
                                                "chip's SIMDs at the guide's clock — the sweeps' floor when every CU holds four groups"}
            except Exception as e:      # (a diagnostic: never the reason a bench line is missing)
                model["issue_model"] = {"error": str(e)}
            if phases:
                sweep_us = max(launch_us - phases["setup_prestep_writeback_us"], 0.0)
                cyc0 = sweep_us * 1e-6 * SHADER_CLOCK_HZ / max(ncol_max * st.impulse_iterations, 1)
                cyc = cyc0
                model.update({"zero_iteration_launch_us": phases["setup_prestep_writeback_us"],
                              "cycles_per_colour_step_by_zero_iteration_subtraction": cyc0,
                              "zero_iteration_note": "launch minus a 0-iteration launch, the figure rounds 1-2 reported; it flatters the class step: "
                                                     "a 0-iteration launch has every workgroup writing back at once and (in-kernel verification) waiting "
                                                     "for the last arrival, neither of which the full launch pays"})
                if "half_sweeps_launch_us" in phases:
                    dsw = st.impulse_iterations - phases["half_sweeps"]
                    step_us = (launch_us - phases["half_sweeps_launch_us"]) / max(ncol_max * dsw, 1)
                    cyc = step_us * 1e-6 * SHADER_CLOCK_HZ
                    sweep_us = step_us * steps_crit
                    model.update({"half_sweeps": phases["half_sweeps"], "half_sweeps_launch_us": phases["half_sweeps_launch_us"],
                                  "how": "slope: (launch with %d impulse sweeps - launch with %d) / (%d sweeps x %d classes of the slowest group)"
                                         % (st.impulse_iterations, phases["half_sweeps"], dsw, ncol_max)})
                model.update({"sweeps_us": sweep_us, "setup_prestep_writeback_us": max(launch_us - sweep_us, 0.0),
                              "achieved_cycles_per_colour_step": cyc, "frac_of_latency_floor": COLOUR_STEP_FLOOR_CYCLES / cyc if cyc > 0 else None,
                              "frac_of_dependent_chain_floor_round3": COLOUR_STEP_CHAIN_FLOOR_CYCLES / cyc if cyc > 0 else None,
                              "setup_writeback_GBps": (tbytes / (max(launch_us - sweep_us, 1e-3) * 1e-6) / 1e9) if tbytes else None})
            if "issue_model" in model:
                model["issue_model"]["status"] = ("a diagnostic, NOT the bound (round 6): one rank of eight — an eighth of the instructions — launches hardly faster "
                                                  "(roofline.chain.one_rank_of_8_launch_us), so the chain of class steps bounds the sweeps; cycles_per_issue 4.5 is this "
                                                  "builder's own probe of a wave64 on a SIMD-16-wide fp32 pipe (profiles/r04_unit_issue_probe.txt: 4.2 plain), the guide's "
                                                  "'2 cycles per VALU instruction' is its figure for packed / SIMD-32 issue")
            roof["latency_model"] = model
            # what bounds the dominant kernel: the dependent chain of class steps of the slowest group (the live half-sweep slope)
            if phases and "half_sweeps_launch_us" in phases and model.get("achieved_cycles_per_colour_step"):
                cyc = model["achieved_cycles_per_colour_step"]
                roof["chain"] = {"class_steps": int(ncol_max * st.impulse_iterations), "classes_of_the_slowest_group": ncol_max, "sweeps": int(st.impulse_iterations),
                                 "us_per_step": cyc / SHADER_CLOCK_HZ * 1e6, "cycles_per_step": cyc,
                                 "sweeps_us": model["sweeps_us"], "launch_us": launch_us, "chain_share_of_launch": model["sweeps_us"] / launch_us if launch_us else None,
                                 "lone_wave_floor_cycles": COLOUR_STEP_FLOOR_CYCLES,
                                 "lone_wave_floor_what": "one wave alone on its SIMD: LDS read 64 + %d VALU x 4 cycles of issue + LDS write 13 + s_barrier 128" % COLOUR_STEP_VALU_INSTRUCTIONS,
                                 "frac_of_lone_wave_floor": COLOUR_STEP_FLOOR_CYCLES / cyc if cyc > 0 else None,
                                 "one_rank_of_8_launch_us": rank_of_8.get("island_launch_us") if rank_of_8 else None,
                                 "one_rank_of_8": rank_of_8,
                                 "what": "the launch is bound by the DEPENDENT CHAIN of class steps of its slowest group (classes x sweeps, one barrier-separated "
                                         "step each), measured as a slope between two sweep counts; measured HBM traffic (roofline.frac) and instruction issue "
                                         "(latency_model.issue_model) are both far from their limits"}
        out = {
            "metric": "solver joint-visits/s (contacts/sec) on the 200k-box stack scene; solver iterations/s in extra",
            "value": visits_all / elapsed_max,
            "unit": "joint-visits/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("cfg2: stack(%d,%d) = %d bodies / %d joints, Single Sloppy island mode, %d+%d iterations, one SolveJoints "
                                    "per step on HBM-resident inputs (bodies in the resident structure-of-arrays layout), schedule cached "
                                    "(built in the warm-up) and checked against the arrays inside the island kernel in every step"
                                    % (args.columns, args.rows, nb, nj, args.iters, args.iters)) if world == 1 else
                                   (("cfg3, replica mode: the same stack(%d,%d) = %d bodies / %d joints world on every rank, Multiple island mode, the "
                                     "schedule's groups dealt to the %d GPUs by joint count, %d+%d iterations, one SolveJoints per step on HBM-resident "
                                     "inputs + pack / all-gather (RCCL) / unpack of the solved bodies and joint impulses every step"
                                     % (args.columns, args.rows, nb_total, nj_total, world, args.iters, args.iters)) if mode == "replica" else
                                    ("cfg3, slab mode (ownership sharding): stack(%d,%d) = %d bodies / %d joints cut into %d x-slabs of whole columns "
                                     "(= islands), one world per GPU holding its slab + the static ground, Multiple island mode, %d+%d iterations, one "
                                     "SolveJoints per step on HBM-resident inputs; the per-step collective is a 4-byte all-reduce (RCCL) queued on the "
                                     "solver's stream — nothing else crosses xGMI" % (args.columns, args.rows, nb_total, nj_total, world, args.iters, args.iters))),
                       "mode": mode, "transport": getattr(group, "backend", "none") if world > 1 else "none",
                       "transport_info": group.info() if hasattr(group, "info") else None,
                       "bodies_total": int(nb_total), "joints_total": int(nj_total), "colours": st.colour_count,
                       "lds_islands": st.lds_islands, "graph_replay": st.graph_replay,
                       "impulse_sweeps_per_step": st.impulse_iterations, "displacement_sweeps_per_step": st.displacement_iterations,
                       "parallelism": (("islands sharded by schedule group, 1 rank per GPU, one all-gather of %d bytes per rank per step"
                                        % solver.exchange_segment_bytes()) if mode == "replica" else
                                       ("islands sharded by x-slab, 1 rank per GPU, one 4-byte all-reduce per step" if mode == "slab" else "1 GPU")),
                       "arith": {0: "source", 1: "fused"}.get(int(phyx_amd._lib.load().phx_arith_mode()), "?") + " (include/phyx_amd.h phx_arith_mode: the sweeps' multiply-add pairs; the oracle has the matching form)",
                       "deviation": static_tag_deviation_note(),
                       "timed_blocks": args.repeats, "reported_block": "median",
                       "device": info["name"], "compute_units": info["compute_units"]},
            "extra": {"solver_iterations_per_sec": iters_max / elapsed_max,
                      "contacts_per_sec_of_solve_time": nj_total * args.steps / elapsed_max,
                      "contacts_resolved_per_sec": None,      # (joints / full World::Update time, SURVEY.md §8(d): filled from cfg2_world_step below)
                      "device_ms_per_step": main_tot["total_ms"] / max(args.steps, 1),
                      "sweep_ms_per_step": main_tot["sweep_ms"] * main_tot["launches"] / max(main_tot["bracketed"], 1) / max(args.steps, 1),
                      "all_blocks_ms_per_step": [round(x, 5) for x in main_tot["all_blocks_ms_per_step"]],
                      "joint_visits_per_sec_sweeps_only": main_tot["visits"] / (main_tot["sweep_ms"] * main_tot["launches"] / max(main_tot["bracketed"], 1) * 1e-3) if main_tot["sweep_ms"] > 0 else None},
            "roofline": roof,
            "result_check": main_tot["result_check"],
        }
        if unbracketed is not None:
            out["extra"]["ms_per_step_without_event_brackets"] = 1e3 * unbracketed["elapsed_max"] / max(args.steps, 1)
            out["extra"]["event_brackets"] = ("ms_per_step / value are measured WITH the HIP-event pairs that bracket the island launch of every 4th "
                                              "step (the live launch time of `roofline`); the same block without any event is the figure above")
        if weak is not None:
            out["extra"]["weak_scaled_slabs"] = weak
        if other_mode is not None:
            out["extra"]["%s_mode" % other_mode["mode"]] = other_mode
        if mode == "replica":
            out["extra"]["exchange"] = {"segment_bytes_per_rank": solver.exchange_segment_bytes(), "status": exchange_status,
                                        "what": "6 floats per body + 2 per joint of the rank's groups behind a 32-byte header {serial, status, "
                                                "topology fingerprint}; status 0 = every rank saw consistent peers in every step"}
        if live_tot is not None:
            out["live_topology"] = {
                "what": "same workload, the schedule (connected components, binning, colouring) rebuilt inside the timed region on EVERY "
                        "solve, as the reference rebuilds PrepareIndices / GatherIslands every call — what a world whose contact graph "
                        "changes every step pays",
                "ms_per_step": 1e3 * live_tot["elapsed_max"] / max(args.steps, 1),
                "joint_visits_per_sec": live_tot["visits"] / live_tot["elapsed_max"]}
        if live_tot is not None:
            out["extra"]["live_topology"] = "see the top-level field"
        if single_tot is not None:
            k = max(5, args.steps // 2)
            sst = single_tot["stats"]
            sl = max(single_tot["launches"], 1)
            s_us = 1e3 * single_tot["sweep_ms"] / max(single_tot["bracketed"], 1)
            s_alg = algorithmic_bytes(single_tot, k) / sl
            skey = "k_solve_dataflow" if sl <= 2 * k else "k_solve_colour"
            s_tr = traffic.get(skey)
            out["extra"]["single_mode"] = "see the top-level field"
            out["single_mode"] = {
                "what": "same input, island_mode = Single (no island split: one coupled system, the path every island too big for a workgroup takes)",
                "ms_per_step": 1e3 * single_tot["elapsed_max"] / k, "joint_visits_per_sec": single_tot["visits"] / single_tot["elapsed_max"],
                "colours": sst.colour_count, "impulse_sweeps_per_step": sst.impulse_iterations,
                "roofline": {"bound": "hbm", "kernel": skey, "launches": single_tot["launches"], "avg_launch_us": s_us,
                             "achieved": s_alg / (s_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": s_alg / (s_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": s_alg,
                             "traffic": s_tr, "traffic_frac": (s_tr / (s_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if s_tr else None}}
        # what a RUNNING cfg 2 world pays, next to `value` (which is the solve on a cached, in-kernel-verified schedule — the one state a
        # running world's changing contact graph never reaches): the solve with the schedule rebuilt every step, and the whole World::Update
        out["live_topology_ms_per_step"] = out["live_topology"]["ms_per_step"] if live_tot is not None else None
        out["world_step_ms_per_step"] = None
        out["contacts_resolved_per_sec"] = None
        if world == 1 and side >= 1:
            ws = cfg2_world_step(world_obj, cfg)
            out["cfg2_world_step"] = ws
            out["world_step_ms_per_step"] = ws["ms_per_step"]
            if (args.columns, args.rows) == (1000, 200):
                out["contacts_resolved_per_sec"] = ws["counts"]["joints"] / (1e-3 * ws["ms_per_step"])
                out["extra"]["contacts_resolved_per_sec"] = out["contacts_resolved_per_sec"]
        if world == 1 and side >= 2:
            out["extra"]["cfg3_one_rank_of_n"] = one_rank_of_n(phyx_amd, Configuration, group, solver, d_bodies, d_cps, d_joints, args, nb, nj, run)
            out["extra"]["cfg3_slab_one_rank_of_n"] = slab_one_rank_of_n(phyx_amd, scenes, Configuration, pdist, device, args, full_scene)
            out["extra"]["other_configs"] = other_configs(phyx_amd, scenes, Configuration, device, world_obj, cfg)
            if (args.columns, args.rows) == (1000, 200):
                out["extra"]["four_times_the_world_one_rank_of_n"] = four_times_the_world_one_rank_of_n(phyx_amd, scenes, Configuration, group, device, args)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(bodies, cps, joints, args.iters, args.cpu_seconds)
        print(json.dumps(out))
    group.shutdown()


def measure_other_mode(phyx_amd, scenes, Configuration, pdist, group, device, args, cfg, mode, full_scene):
    """The sharding mode the headline did not use, measured the same way (median of 3 blocks): returns its ms per step and
    whole-job joint-visits/s."""
    rank, world = group.rank, group.world_size
    scene = pdist.slab_partition(full_scene, world)[rank][0] if mode == "slab" else full_scene
    w = phyx_amd.World(device, gravity=-200.0)
    w.add_scene(scene)
    for _ in range(args.scene_steps):
        w.Update(1.0 / 60.0, cfg)
    w.PreSolve(1.0 / 60.0)
    arrs = [phyx_amd.DeviceArray(a, device) for a in (w.bodies, w.contactPoints, w.contactJoints)]
    slv = phyx_amd.Solver(device)
    xch = None
    if mode == "replica":
        slv.set_shard(rank, world)
        xch = group.exchange(slv, pdist.Exchange.capacity_for(arrs[0].count, arrs[2].count))
    hook = xch.hook() if xch else group.stream_hook(slv.stream_ptr())
    for _ in range(2):
        slv.bench(arrs[0], arrs[1], arrs[2], cfg, 0, 1, hook=hook)
    times, visits = [], 0
    for _ in range(3):
        slv.bench_stage(arrs[0], arrs[2], args.steps)
        group.barrier(); slv.synchronize()
        t0 = time.perf_counter()
        r = slv.bench(arrs[0], arrs[1], arrs[2], cfg, 0, args.steps, hook=hook)
        slv.synchronize(); group.barrier()
        times.append(group.reduce_max(time.perf_counter() - t0))
        visits = group.reduce_sum(r.joint_visits)
    el = sorted(times)[1]
    return {"mode": mode, "ms_per_step": 1e3 * el / max(args.steps, 1), "joint_visits_per_sec": visits / el, "scaling": "strong",
            "exchange_status": int(group.reduce_max(slv.exchange_status())) if xch else 0}


def world_groups_max_colours(solver):
    """Colours of the schedule group with the most colours (the critical path of the island launch)."""
    _, offs = solver.schedule()
    groups, lds = solver.groups()
    offs = np.asarray(offs)
    # colour_offsets are slot offsets; a group's colours = the colour boundaries that fall inside its slot range
    best = 0
    idx = np.searchsorted(offs, np.asarray(groups))
    for g in range(lds):
        best = max(best, int(idx[g + 1] - idx[g]))
    return best


def one_rank_of_n(phyx_amd, Configuration, group, solver, d_bodies, d_cps, d_joints, args, nb, nj, run, ns=(1, 2, 4, 8)):
    """What ONE rank of an n-GPU config-3 run spends per step, measured on this GPU: shard 0 of n of the Multiple-mode
    schedule + pack + (the all-gather replaced by a local copy of the rank's own segment) + unpack.  The RCCL time is not in
    it; it bounds the strong scaling the driver's multi-GPU run can show."""
    from phyx_amd import dist as pdist
    cfg3 = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE, args.iters, args.iters)
    res = {}
    slv = phyx_amd.Solver(solver.device)
    xch = pdist.Exchange(group, slv, pdist.Exchange.capacity_for(nb, nj), solver.device)
    for n in ns:
        slv.set_shard(0, n)
        tot = run(cfg3, 2, 10, 3, slv=slv, hk=xch.hook())
        res["n=%d" % n] = {"ms_per_step": 1e3 * tot["elapsed_max"] / 10, "island_launch_us": 1e3 * tot["sweep_ms"] / max(tot["bracketed"], 1),
                           "segment_bytes": slv.exchange_segment_bytes(), "groups": (tot["stats"].lds_islands + n - 1) // n}
    return res


def slab_one_rank_of_n(phyx_amd, scenes, Configuration, pdist, device, args, full_scene):
    """What ONE rank of an n-GPU config-3 run in SLAB mode (ownership sharding: bench.py --gpus n, the default) spends per step,
    measured on this GPU: the solver on rank 0's x-slab of the 200k-box world, no collective (a 4-byte all-reduce in the real
    run).  A rank with 1/n of the islands is NOT n times faster: the island launch lasts as long as one island's ~84 dependent
    class steps however few islands there are."""
    cfg3 = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE, args.iters, args.iters)
    res = {}
    for n in (2, 4, 8):
        sub = pdist.slab_partition(full_scene, n)[0][0]
        w = phyx_amd.World(device, gravity=-200.0)
        w.add_scene(sub)
        for _ in range(args.scene_steps):
            w.Update(1.0 / 60.0, cfg3)
        w.PreSolve(1.0 / 60.0)
        arrs = [phyx_amd.DeviceArray(a, device) for a in (w.bodies, w.contactPoints, w.contactJoints)]
        del w
        slv = phyx_amd.Solver(device)
        for _ in range(2):
            slv.bench(arrs[0], arrs[1], arrs[2], cfg3, 0, 1)
        best = None
        for _ in range(3):
            slv.bench_stage(arrs[0], arrs[2], 10)
            slv.synchronize()
            t0 = time.perf_counter()
            r = slv.bench(arrs[0], arrs[1], arrs[2], cfg3, 0, 10)
            slv.synchronize()
            el = (time.perf_counter() - t0) / 10
            best = el if best is None else min(best, el)
        res["n=%d" % n] = {"ms_per_step": 1e3 * best, "island_launch_us": 1e3 * r.impulse_kernel_ms / max(r.bracketed_launches, 1),
                           "joints": arrs[2].count, "groups": slv.stats().lds_islands}
    return res


def four_times_the_world_one_rank_of_n(phyx_amd, scenes, Configuration, group, device, args):
    """Where island sharding does scale: a world of 4000 columns (800k boxes) is four residency rounds of the island kernel on one
    GPU (1024 workgroups are resident at once); with 1/N of the groups a rank needs 4/N rounds.  Measured like
    cfg3_one_rank_of_n: shard 0 of N on this GPU, pack + local copy + unpack, no RCCL time."""
    import time
    from phyx_amd import dist as pdist
    cfg3 = Configuration(phyx_amd.SOLVE_AVX2, phyx_amd.ISLAND_MULTIPLE, args.iters, args.iters)
    w = phyx_amd.World(device, gravity=-200.0)
    w.add_scene(scenes.stack(4 * args.columns, args.rows))
    for _ in range(args.scene_steps):
        w.Update(1.0 / 60.0, cfg3)
    w.PreSolve(1.0 / 60.0)
    arrs = [phyx_amd.DeviceArray(a, device) for a in (w.bodies, w.contactPoints, w.contactJoints)]
    nb, nj = arrs[0].count, arrs[2].count
    del w
    slv = phyx_amd.Solver(device)
    xch = pdist.Exchange(group, slv, pdist.Exchange.capacity_for(nb, nj), device)
    res = {"bodies": nb, "joints": nj}
    for n in (1, 2, 4, 8):
        slv.set_shard(0, n)
        slv.bench(arrs[0], arrs[1], arrs[2], cfg3, 0, 1, hook=xch.hook())
        best = None
        for _ in range(3):
            slv.bench_stage(arrs[0], arrs[2], 5)
            slv.synchronize()
            t0 = time.perf_counter()
            r = slv.bench(arrs[0], arrs[1], arrs[2], cfg3, 0, 5, hook=xch.hook())
            slv.synchronize()
            el = (time.perf_counter() - t0) / 5
            best = el if best is None else min(best, el)
        res["n=%d" % n] = {"ms_per_step": 1e3 * best, "island_launch_us": 1e3 * r.impulse_kernel_ms / max(r.bracketed_launches, 1),
                           "groups": (slv.stats().lds_islands + n - 1) // n}
    return res


def cfg2_world_step(cfg2_world, cfg2):
    """The whole device-resident World::Update of the bench's own 200k-box world (ref: World.cpp:19-37), topology still changing (new
    contacts every step): median of 9 synchronised Update calls + one step with per-phase host timers."""
    # (the world arrives behind a PreSolve — the solver bench ran on that step's joints: finish that step untimed, time whole Update
    #  calls as a caller of the reference's World::Update makes them, and leave the world behind a PreSolve again)
    cfg2_world.FinishStep(1.0 / 60.0, cfg2); cfg2_world.sync()
    t = []
    for _ in range(9):
        t0 = time.perf_counter(); cfg2_world.Update(1.0 / 60.0, cfg2); cfg2_world.sync()
        t.append(time.perf_counter() - t0)
    cfg2_world.PreSolve(1.0 / 60.0); cfg2_world.sync()
    cfg2_world.set_phase_timing(True)               # the per-phase breakdown costs a synchronisation per phase: measured separately
    cfg2_world.FinishStep(1.0 / 60.0, cfg2); cfg2_world.PreSolve(1.0 / 60.0)
    ph = cfg2_world.phase_ms()
    cfg2_world.set_phase_timing(False)
    return {"what": "World::Update (IntegrateVelocity, UpdateBroadphase, UpdatePairs, UpdateManifolds, PackManifolds, RefreshContactJoints, SolveJoints with "
                    "the schedule rebuilt, IntegratePosition) of the 200k-box world, synchronised per step",
            "ms_per_step": 1e3 * float(np.median(t)), "phases_ms": {k: round(v, 3) for k, v in ph.items()},
            "counts": dict(zip(("bodies", "manifolds", "contact_points", "joints"), cfg2_world.counts()))}


def other_configs(phyx_amd, scenes, Configuration, device, cfg2_world, cfg2):
    """Untimed-by-the-driver side measurements of the other BASELINE.json configs (N=1, --secondary): the settled 200k-box world,
    the broadphase at cfg 4 size (1M boxes) and the solver at cfg 5 size (500k boxes, 50 iterations).  They are parity-test cases
    first (tests/test_*_gpu.py); the numbers here are informational."""
    res = {}
    # the same world once it has settled: around step 30 the columns' islands merge into ONE island of ~7e5 joints that no workgroup
    # holds — the steady state a user of a long-running stack sees, solved class by class out of HBM
    cfg2_world.FinishStep(1.0 / 60.0, cfg2)                           # (the loop above left the world behind a PreSolve)
    for _ in range(42):                                              # ~15 steps so far
        cfg2_world.Update(1.0 / 60.0, cfg2)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); cfg2_world.Update(1.0 / 60.0, cfg2); cfg2_world.sync(); t.append(time.perf_counter() - t0)
    sst = cfg2_world.solver.stats()
    ki, parts, launches = cfg2_world.solver.partition()
    res["settled_world_step"] = {"what": "World::Update of the same 200k-box world at steps 57-61: the columns have merged into one island (HBM path; its "
                                         "interior units — both bodies in one block of 512 body indices, of the plain grid or of the grid shifted by 256 — "
                                         "swept by one launch per level and sweep, a workgroup per block)",
                                 "ms_per_step": 1e3 * float(np.median(t)), "lds_islands": sst.lds_islands, "colours": sst.colour_count,
                                 "interior_classes": ki, "parts": parts, "sweep_launches_per_solve": launches,
                                 "counts": dict(zip(("bodies", "manifolds", "contact_points", "joints"), cfg2_world.counts()))}
    # ... and the same world 40 steps later: the pile loosens again (hundreds of loose pieces beside the big island, whose boundary units — bodies
    # whose neighbours lie far away in the body order — need up to ten small trailing classes; k_solve_tail sweeps those in one launch)
    for _ in range(38):
        cfg2_world.Update(1.0 / 60.0, cfg2)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); cfg2_world.Update(1.0 / 60.0, cfg2); cfg2_world.sync(); t.append(time.perf_counter() - t0)
    sst = cfg2_world.solver.stats()
    ki, parts, launches = cfg2_world.solver.partition()
    res["loosened_world_step"] = {"what": "World::Update of the same world at steps 100-104: the big island beside hundreds of workgroup-sized ones",
                                  "ms_per_step": 1e3 * float(np.median(t)), "lds_islands": sst.lds_islands, "colours": sst.colour_count,
                                  "interior_classes": ki, "parts": parts, "sweep_launches_per_solve": launches,
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
    # algorithmic bytes: 112 B per body for key build + radix sort + gather, 20 B per candidate test.  The sweep
    # is an L1-resident latency loop (PMC: ~6 % of its algorithmic bytes reach HBM), so this is NOT quoted against the HBM peak.
    alg = 112.0 * w4.counts()[0] + 20.0 * bs.candidate_tests
    # measured traffic: ONLY the kernels of one profiled steady update, each weighted by its launches in that update and listed with its
    # time (tools/steady_step_summary.py: kernel trace + the two PMC passes of the same step) — round 4 summed every kernel name in
    # the PMC file, one-off first-update kernels included, and its 'traffic = algorithmic' was a coincidence
    pm4 = steady_step_file("r06_cfg4_steady_step") or steady_step_file("r05_cfg4_steady_step")
    bp_rows = [r for r in pm4.get("kernels", []) if r.get("phase") == "broadphase"]
    bp_traffic = sum(r["hbm_bytes"] for r in bp_rows) or None
    bp_us = sum(r["us"] for r in bp_rows) or None
    res["cfg4_broadphase_1M"] = {"device_ms": bs.device_ms, "candidate_tests": bs.candidate_tests, "new_pairs": bs.new_pairs,
                                 "candidate_tests_per_sec": bs.candidate_tests / (bs.device_ms * 1e-3),
                                 "algorithmic_GBps_cache_resident": alg / (bs.device_ms * 1e-3) / 1e9, "world_step_ms": 1e3 * step4,
                                 "counts": dict(zip(("bodies", "manifolds", "contact_points", "joints"), w4.counts())),
                                 "roofline": {"bound": "hbm", "kernel": "UpdateBroadphase + UpdatePairs (key build, 3-pass radix sort, gather, sweep k_sweep_rows)",
                                              "algorithmic_bytes_per_update": alg, "what": "SURVEY.md §8(d): 112 B per body + 20 B per candidate test",
                                              "achieved": alg / (bs.device_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": alg / (bs.device_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "traffic": bp_traffic, "traffic_source": pm4.get("file"),
                                              "traffic_frac": (bp_traffic / (bp_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if bp_traffic and bp_us else None,
                                              "traffic_frac_what": "measured HBM bytes of the broadphase kernels of one profiled steady update over the sum of "
                                                                   "their kernel times in that update (both from the file; per kernel below)",
                                              "per_kernel": [{"kernel": r["kernel"], "launches": r["launches"], "us": r["us"], "hbm_bytes": r["hbm_bytes"]} for r in bp_rows] or None,
                                              "note": "the sweep re-reads neighbouring entries out of L1/L2 (one 20-byte entry serves ~50 tests): measured HBM "
                                                      "traffic is a small fraction of the no-reuse algorithmic figure, and the update is dispatch- and latency-bound"}}
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
    launch5_us = 1e3 * r.impulse_kernel_ms / max(r.bracketed_launches, 1)
    pm5 = pmc_file("r06_pmc_traffic_cfg5") or pmc_file("r05_pmc_traffic_cfg5")
    tr5 = next((v for k, v in pm5.get("kernels", {}).items() if "k_solve_islands<512" in k), None)
    alg5 = (BYTES_IMPULSE_VISIT * r.joint_visits + BYTES_DISPLACEMENT_VISIT * st.displacement_iterations * arrs[2].count * 10) / max(r.impulse_launches, 1)
    res["cfg5_500k_tall_50it_fp32"] = {"ms_per_step": 1e3 * el / 10, "joint_visits_per_sec": r.joint_visits / el, "joints": arrs[2].count,
                                       "impulse_sweeps": st.impulse_iterations, "lds_islands": st.lds_islands, "sweep_ms_per_step": r.impulse_kernel_ms * r.impulse_launches / max(r.bracketed_launches, 1) / 10,
                                       "roofline": {"bound": "hbm", "kernel": "k_solve_islands<512,1024> (two residency rounds of 512 workgroups)", "avg_launch_us": launch5_us,
                                                    "traffic": tr5, "traffic_source": pm5.get("file"),
                                                    "achieved": (tr5 / (launch5_us * 1e-6) / 1e9) if tr5 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                    "frac": (tr5 / (launch5_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if tr5 else None,
                                                    "algorithmic_bytes_per_launch": alg5, "algorithmic_GBps": alg5 / (launch5_us * 1e-6) / 1e9}}
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
        "sweep_ms_per_step": r16.impulse_kernel_ms * r16.impulse_launches / max(r16.bracketed_launches, 1) / 10,
        "max_abs_velocity_diff_vs_fp32": float(max(np.abs(b16["velocity"]["x"] - b32["velocity"]["x"]).max(), np.abs(b16["velocity"]["y"] - b32["velocity"]["y"]).max())),
        "mean_abs_velocity_diff_vs_fp32": float(np.abs(b16["velocity"]["y"] - b32["velocity"]["y"]).mean()),
        "max_abs_impulse_diff_vs_fp32": float(np.abs(j16["normal_acc"] - j32["normal_acc"]).max())}
    return res


def static_tag_deviation_note():
    """What 'bit-exact' is quoted next to: the device's static-tag rule (a private, class-synchronous copy per group; every
    group leaves its sweeps on its own) against the reference's Single-mode rule (one shared word per static body, one early exit) on the
    device's own order, measured by tests/test_solver_gpu.py::test_static_tag_rule_deviation_at_full_size and kept in profiles/."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r06_static_tag_deviation.json")))["cfg2"]
        return {"what": "results are bit-exact against the oracle replaying the device's schedule under the device's static-tag rule; against the "
                        "reference's Single-mode rule on the same order (one shared lastIteration word per static body, one early exit) the solve "
                        "differs as stated — outside SURVEY §8(c) T1 (1e-3 in a velocity), inside this backend's stated tolerance (5e-3 in an impulse, 5 in a velocity, 0.1 in a position after the step)",
                "bodies_differing": d["bodies_differing"], "bodies": d["bodies"], "max_abs_dvel": d["max_abs_dvel"], "max_abs_dimpulse": d["max_abs_dimpulse"],
                "max_abs_dpos_after_integrate": d["max_abs_dpos_after_integrate"], "stag_events": d["stag_events"], "inside_T1": d["inside_T1"],
                "source": "profiles/r06_static_tag_deviation.json (tools/static_tag_deviation.py; NOT measured in this run)"}
    except Exception:
        return None


def physical_cores():
    """cores among the CPUs this process may use: CPUs that lead their thread_siblings_list (oracle/cpu_baseline.c pool_cpus orders its workers the same way)"""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1
    n = 0
    for cpu in allowed:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu) as f:
                first = int(f.read().replace("-", ",").split(",")[0])
        except (OSError, ValueError):
            first = cpu
        n += 1 if first == cpu else 0
    return max(n, 1)


def cpu_baseline(bodies, cps, joints, iters, budget_s):
    """The host-CPU baseline beside the GPU number, timed in this run on this host: oracle/cpu_baseline.c, an
    8-wide AVX2-order restatement of Solver::SolveJoints<8> (greedy 8-grouping ref: Solver.cpp:217-273, group-granular skip
    :797, ContactJointPacked<8> layout) built with the reference's own flags (-O3 -ffast-math -mavx2 -mfma) — its strict build
    is bit-equal to the oracle's AVX2 mode (tests/test_oracle_solver.py).  One full SolveJoints of the same solver input per
    sample, phases timed around the reference's scopes; 1 thread = island mode Single, T threads = Single Sloppy (512-joint
    batches over persistent threads, ref: Solver.cpp:138-139).  T = best of a probe (the racy sweep stops scaling long before
    all cores).  `value` = impulse-loop joint-visits/s at T threads; the scalar (N = 1) oracle loop is kept next to it.
    Reported beside the GPU number, not a target."""
    from oracle import binding as ob
    try:
        ncpu = len(os.sched_getaffinity(0))            # the cores this process may really use (a container's share), not the box's
    except AttributeError:
        ncpu = os.cpu_count() or 1
    cores = physical_cores()
    b = bodies.view(ob.body_dtype); cp = cps.view(ob.contact_point_dtype); j = joints.view(ob.joint_dtype)
    t_begin = time.perf_counter()
    names = ("prepare_bodies", "prepare_indices", "prepare_joints", "refresh", "prestep", "impulse", "displacement", "finish", "total")

    def solve(threads):
        ph = ob.baseline_solve(b.copy(), cp, j.copy(), iters, iters, threads, threads > 1, "fast")
        return {n: getattr(ph, n) for n in names}, ph.joint_visits, ph.impulse_iterations

    # `value` is taken on ALL host threads; a short probe over smaller counts rides along, because the racy
    # 512-joint sweep stops scaling long before all cores and the best count says so
    probe = {}
    for t in sorted({1, 4, 8, 16, 32, 64, 128, cores, ncpu}):
        if t > ncpu or time.perf_counter() - t_begin > 0.5 * budget_s:
            continue
        solve(t)
        ph, visits, _ = solve(t)
        probe[t] = visits / ph["impulse"]
    best = max(probe, key=probe.get) if probe else ncpu

    def sample(threads, seconds):
        acc = {n: [] for n in names}
        visits = sweeps = 0
        t0 = time.perf_counter()
        while True:
            ph, visits, sweeps = solve(threads)
            for n in names:
                acc[n].append(ph[n])
            if time.perf_counter() - t0 > seconds or len(acc["total"]) >= 50:
                break
        med = {n: float(np.median(v)) for n, v in acc.items()}
        return med, visits, sweeps, len(acc["total"])

    left = max(budget_s - (time.perf_counter() - t_begin), 2.0)
    one, v1, sw1, n1 = sample(1, 0.25 * left)
    # every host thread AND the probe's best count: 512-joint batches behind one shared counter stop scaling
    # long before 256 threads (the reference's parallelFor has the same shape, ref: base/Parallel.h:44-77); `value` is the better of
    # the two samples, `cores` the thread count it was taken with, and the all-threads figure is reported next to it either way
    many, vm, swm, nm = sample(ncpu, 0.25 * left)
    all_threads = {"threads": ncpu, "value": vm / many["impulse"], "solve_ms_per_step": 1e3 * many["total"], "samples": nm}
    used = ncpu
    if best != ncpu:
        m2, v2, s2, n2 = sample(best, 0.25 * left)
        if v2 / m2["impulse"] > vm / many["impulse"]:
            many, vm, swm, nm, used = m2, v2, s2, n2, best
    best = used
    # broadphase phases (UpdateBroadphase is serial in the reference even with workers; UpdatePairs in blocks of 128 rows)
    bp1 = ob.baseline_broadphase(b, 1, 3, "fast")
    bpm = ob.baseline_broadphase(b, best, 3, "fast")
    # the scalar (N = 1) restatement, impulse loop only (the round-1 baseline)
    sec, v = ob.time_impulse_loop(b, cp, j, iters, 1)
    ms = lambda d: {k: round(1e3 * x, 3) for k, x in d.items()}
    return {"value": vm / many["impulse"], "unit": "joint-visits/s", "cores": best, "host_threads": ncpu, "host_cores": cores, "kind": "port",
            "threads": best, "cores_used": min(best, cores),
            "pinning": "worker w pinned to the w-th allowed CPU, one hardware thread of every core first (up to %d workers sit on different cores); every worker "
                       "sweeps the same contiguous share of the 512-joint batches in every phase — the packs it touched first — and steals from the "
                       "others' shares only when it has finished its own" % cores,
            "probe_M_visits_per_s_by_threads": {str(k): round(x / 1e6) for k, x in sorted(probe.items())},
            "all_host_threads": all_threads,
            "sample": "%d x one full SolveJoints<8> of the same %d-joint solver input (%d impulse sweeps run), AVX2-order baseline built "
                      "-O3 -ffast-math -mavx2 -mfma, %d pinned threads (the better of: every host thread / the probe's best count), 512-joint batches "
                      "in contiguous per-worker shares with stealing (Single Sloppy; probe %s M visits/s by thread count; %d host threads); value = joints x "
                      "sweeps / median impulse-loop time" % (nm, len(j), swm, best, {k: round(x / 1e6) for k, x in probe.items()}, ncpu),
            "solve_ms_per_step": 1e3 * many["total"], "phases_ms": ms(many),
            "single_thread": {"value": v1 / one["impulse"], "solve_ms_per_step": 1e3 * one["total"], "phases_ms": ms(one), "samples": n1,
                              "island_mode": "Single"},
            "broadphase_ms": {"threads_1": {"UpdateBroadphase": 1e3 * bp1.update_broadphase / 3, "UpdatePairs": 1e3 * bp1.update_pairs / 3},
                              "threads_%d" % best: {"UpdateBroadphase": 1e3 * bpm.update_broadphase / 3, "UpdatePairs": 1e3 * bpm.update_pairs / 3},
                              "candidate_tests": int(bp1.candidate_tests), "what": "steady state: every pair already in the persistent set (lookups only)"},
            "scalar_port_single_thread_value": v / sec,
            "calibration": "the reference's own build cannot run here (un-vendored microprofile.h); BASELINE.md §2's survey probe of the compiled "
                           "reference on an 8-vCPU Xeon gives 179 M joint-visits/s (1 thread, AVX2) and 49.5 ms / 9.1 ms for Impulse / "
                           "RefreshJoints at 443k joints; this baseline measured 178 M/s, 49.7 ms and 8.6 ms (scaled) on that same container"}


if __name__ == "__main__":
    main()
