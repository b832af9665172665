"""Summarise rocprofv3 outputs under gpurun_out/prof into profiles/ (committed evidence).

  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of `python bench.py`
  profiles/<tag>_pmc_traffic.json      FETCH_SIZE / WRITE_SIZE per launch per kernel (separate --pmc passes)

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB
(hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024); on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced streams, so the read side is doubled ("corrected"); both raw and corrected figures are kept, WRITE_SIZE
is uncalibrated and used as is.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def per_kernel(path):
    """Counter sums per kernel.  One kernel is launched with several grids (bench.py's side measurements shard the island
    launch, the world steps have fewer groups): per kernel only the dispatches of its MOST FREQUENT grid size are kept — for the
    island kernel that is the full 999-workgroup launch of the timed region."""
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        rows[r["Kernel_Name"]].append((r["Grid_Size"], float(r["Counter_Value"])))
    agg = {}
    for k, v in rows.items():
        grid = collections.Counter(g for g, _ in v).most_common(1)[0][0]
        kept = [x for g, x in v if g == grid]
        agg[k] = [len(kept), sum(kept), grid]
    return agg


def main(tag, dominant):
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, "trace_kernel_stats.csv"), os.path.join(DST, tag + "_kernel_stats.csv"))
    if os.path.exists(os.path.join(SRC, "main_kernel_stats.csv")):       # the main measurement alone: every island launch is a full one
        shutil.copy(os.path.join(SRC, "main_kernel_stats.csv"), os.path.join(DST, tag + "_kernel_stats_main_only.csv"))
    for name in ("bench_plain.json", "bench_trace.json", "bench_main.json"):
        if os.path.exists(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(DST, tag + "_" + name))
    fetch = per_kernel(os.path.join(SRC, "pmc_fetch_counter_collection.csv"))
    write = per_kernel(os.path.join(SRC, "pmc_write_counter_collection.csv"))
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        nf, vf, grid = fetch.get(k, [0, 0.0, None])
        nw, vw, _ = write.get(k, [0, 0.0, None])
        f = vf / nf * 1024 if nf else 0.0
        w = vw / nw * 1024 if nw else 0.0
        kernels[k] = {"launches": max(nf, nw), "grid_size": grid, "fetch_bytes_per_launch_raw": f, "write_bytes_per_launch_raw": w,
                      "hbm_bytes_per_launch_raw": f + w, "hbm_bytes_per_launch_corrected": 2 * f + w}
    dom = [k for k in kernels if dominant in k]
    import hashlib, subprocess
    h = hashlib.sha256()
    for f in ("island_kernel.h", "solver_kernels.h"):
        h.update(open(os.path.join(ROOT, "phyx_amd", "csrc", f), "rb").read())
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "phyx_amd/csrc"], text=True).strip():
            commit += "+uncommitted csrc changes"
    except Exception:
        commit = "unrecorded"
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline",
           # bench.py refuses this file once island_kernel.h / solver_kernels.h differ from what was profiled (kernel_source_sha256)
           "commit": commit, "kernel_source_sha256": h.hexdigest(),
           "correction": "read side x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md §HBM); WRITE_SIZE as reported",
           "dominant_kernel": dom[0] if dom else None,
           "hbm_bytes_per_launch": kernels[dom[0]]["hbm_bytes_per_launch_corrected"] if dom else None,
           "kernels": kernels}
    # the HBM path's class kernel runs with one grid per class: bench.py's Single-mode measurement is the set of big grids that
    # occur most often (its 4 classes x sweeps x steps); the side measurements' merged islands have hundreds of other grid sizes
    # and the settled world's tiny tail classes share the 64-lane grid, which is the most frequent one overall
    def colour_launches(path):
        per_grid = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if "k_solve_colour<true, true>" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1024:
                per_grid[r["Grid_Size"]].append(float(r["Counter_Value"]))
        if not per_grid:
            return {}
        top = max(len(v) for v in per_grid.values())
        return {g: v for g, v in per_grid.items() if 2 * len(v) >= top}
    cf = colour_launches(os.path.join(SRC, "pmc_fetch_counter_collection.csv"))
    cw = colour_launches(os.path.join(SRC, "pmc_write_counter_collection.csv"))
    for name, k in kernels.items():
        if "k_solve_colour<true, true>" in name and cf:
            nf = sum(len(v) for v in cf.values()); nw = max(sum(len(v) for v in cw.values()), 1)
            f = sum(sum(v) for v in cf.values()) / nf * 1024
            w = sum(sum(v) for v in cw.values()) / nw * 1024
            out["hbm_colour_kernel"] = name
            out["hbm_colour_grids"] = {g: len(v) for g, v in sorted(cf.items(), key=lambda x: int(x[0]))}
            out["hbm_colour_bytes_per_launch"] = 2 * f + w
            out["hbm_colour_bytes_per_launch_by_grid"] = {g: 2 * 1024 * sum(v) / len(v) + (1024 * sum(cw[g]) / len(cw[g]) if g in cw else 0.0)
                                                           for g, v in sorted(cf.items(), key=lambda x: int(x[0]))}
    # duration of the solve kernels over the launches bench.py times: the kernel-trace stats also average the scene's own first
    # world steps (3 launches on a stack that has hardly any contacts yet) and the 0-iteration phase measurement, so recompute
    # from the raw trace: the launches of the main timed region are the ones within 25 % of the median
    trace = os.path.join(SRC, "trace_kernel_trace.csv")
    if os.path.exists(trace):
        rows_all = list(csv.DictReader(open(trace)))
        for key, needle in (("dominant_kernel_launch_us", dominant), ("hbm_colour_kernel_launch_us", "k_solve_colour<true, true>"), ("tail_kernel_launch_us", "k_solve_tail")):
            mine = [r for r in rows_all if needle in r["Kernel_Name"]]
            if not mine:
                continue
            grid = collections.Counter(r["Grid_Size_X"] for r in mine).most_common(1)[0][0]     # the full launch of the timed region
            grids = set(cf) if (needle.startswith("k_solve_colour") and cf) else {grid}
            if grids != {grid}:
                grid = "+".join(sorted(grids, key=int))
            us = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in mine if r["Grid_Size_X"] in grids)
            med = us[len(us) // 2]
            timed = [u for u in us if 0.75 * med <= u <= 1.25 * med]
            out[key] = {"launches": len(us), "grid_size": grid, "median": med, "mean": sum(us) / len(us),
                        "mean_within_25pct_of_median": sum(timed) / len(timed), "min": us[0], "max": us[-1]}
    for extra in ("world_kernel_stats.csv", "world_step_timeline.txt", "settled_last_step.txt", "settled_kernel_stats.csv"):
        if os.path.exists(os.path.join(SRC, extra)):
            shutil.copy(os.path.join(SRC, extra), os.path.join(DST, tag + "_" + extra))
    if os.path.exists(os.path.join(SRC, "world_marker_api_stats.csv")):
        shutil.copy(os.path.join(SRC, "world_marker_api_stats.csv"), os.path.join(DST, tag + "_world_marker_stats.csv"))
    json.dump(out, open(os.path.join(DST, tag + "_pmc_traffic.json"), "w"), indent=1)
    print("dominant:", out["dominant_kernel"], "->", out["hbm_bytes_per_launch"], "B per launch")


def source_stamp():
    """(commit, sha256 of island_kernel.h + solver_kernels.h): bench.py refuses a file taken of another version of the kernels"""
    import hashlib, subprocess
    h = hashlib.sha256()
    for f in ("island_kernel.h", "solver_kernels.h"):
        h.update(open(os.path.join(ROOT, "phyx_amd", "csrc", f), "rb").read())
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "phyx_amd/csrc"], text=True).strip():
            commit += "+uncommitted csrc changes"
    except Exception:
        commit = "unrecorded"
    return commit, h.hexdigest()


def side_config(tag, name, dominant):
    """profiles/<tag>_pmc_traffic_<name>.json + kernel stats + (cfg4) the step timeline of a tools/prof_cfg.py workload"""
    fetch = per_kernel(os.path.join(SRC, name + "_pmc_fetch_counter_collection.csv"))
    write = per_kernel(os.path.join(SRC, name + "_pmc_write_counter_collection.csv"))
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        nf, vf, grid = fetch.get(k, [0, 0.0, None])
        nw, vw, _ = write.get(k, [0, 0.0, None])
        f = vf / nf * 1024 if nf else 0.0
        w = vw / nw * 1024 if nw else 0.0
        kernels[k] = {"launches": max(nf, nw), "grid_size": grid, "fetch_bytes_per_launch_raw": f, "write_bytes_per_launch_raw": w,
                      "hbm_bytes_per_launch_raw": f + w, "hbm_bytes_per_launch_corrected": 2 * f + w}
    dom = [k for k in kernels if dominant in k]
    commit, sha = source_stamp()
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/prof_cfg.py " + name,
           "commit": commit, "kernel_source_sha256": sha,
           "correction": "read side x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md §HBM); WRITE_SIZE as reported",
           "dominant_kernel": dom[0] if dom else None,
           "hbm_bytes_per_launch": kernels[dom[0]]["hbm_bytes_per_launch_corrected"] if dom else None, "kernels": kernels}
    stats = os.path.join(SRC, name + "_trace_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(DST, "%s_%s_kernel_stats.csv" % (tag, name)))
        for r in csv.DictReader(open(stats)):
            if dominant in r["Name"]:
                out["dominant_kernel_avg_us"] = float(r["AverageNs"]) / 1e3
                out["dominant_kernel_GBps"] = out["hbm_bytes_per_launch"] / (float(r["AverageNs"]) * 1e-9) / 1e9 if out["hbm_bytes_per_launch"] else None
    tl = os.path.join(SRC, name + "_step_timeline.txt")
    if os.path.exists(tl):
        shutil.copy(tl, os.path.join(DST, "%s_%s_step_timeline.txt" % (tag, name)))
    json.dump(out, open(os.path.join(DST, "%s_pmc_traffic_%s.json" % (tag, name)), "w"), indent=1)
    print(name, "dominant:", out["dominant_kernel"], "->", out["hbm_bytes_per_launch"], "B per launch", out.get("dominant_kernel_avg_us"), "us")


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    main(tag, sys.argv[2] if len(sys.argv) > 2 else "k_solve_islands<256")
    if os.path.exists(os.path.join(SRC, "cfg4_pmc_fetch_counter_collection.csv")):
        side_config(tag, "cfg4", "k_sweep_rows")
    if os.path.exists(os.path.join(SRC, "cfg5_pmc_fetch_counter_collection.csv")):
        side_config(tag, "cfg5", "k_solve_islands<512")
    if os.path.exists(os.path.join(SRC, "settled_pmc_fetch_counter_collection.csv")):      # tools/gpu_parts_prof.sh N pmc
        side_config(tag, "settled", "k_solve_parts<true, true, false>")
