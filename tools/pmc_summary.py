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
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main(tag, dominant):
    os.makedirs(DST, exist_ok=True)
    shutil.copy(os.path.join(SRC, "trace_kernel_stats.csv"), os.path.join(DST, tag + "_kernel_stats.csv"))
    for name in ("bench_plain.json", "bench_trace.json"):
        if os.path.exists(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(DST, tag + "_" + name))
    fetch = per_kernel(os.path.join(SRC, "pmc_fetch_counter_collection.csv"))
    write = per_kernel(os.path.join(SRC, "pmc_write_counter_collection.csv"))
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(k, [0, 0.0])
        nw, vw = write.get(k, [0, 0.0])
        f = vf / nf * 1024 if nf else 0.0
        w = vw / nw * 1024 if nw else 0.0
        kernels[k] = {"launches": max(nf, nw), "fetch_bytes_per_launch_raw": f, "write_bytes_per_launch_raw": w,
                      "hbm_bytes_per_launch_raw": f + w, "hbm_bytes_per_launch_corrected": 2 * f + w}
    dom = [k for k in kernels if dominant in k]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline",
           "correction": "read side x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md §HBM); WRITE_SIZE as reported",
           "dominant_kernel": dom[0] if dom else None,
           "hbm_bytes_per_launch": kernels[dom[0]]["hbm_bytes_per_launch_corrected"] if dom else None,
           "kernels": kernels}
    # duration of the dominant kernel over the launches bench.py times: the kernel-trace stats also average the scene's own
    # first world steps (3 launches on a stack that has hardly any contacts yet), so recompute from the raw trace
    trace = os.path.join(SRC, "trace_kernel_trace.csv")
    if dom and os.path.exists(trace):
        rows = [r for r in csv.DictReader(open(trace)) if r["Kernel_Name"] == dom[0]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        timed = us[3:]                                  # bench.py --scene-steps 3
        out["dominant_kernel_launch_us"] = {"all_launches": us, "scene_steps_dropped": 3, "mean_of_the_rest": sum(timed) / max(len(timed), 1),
                                            "min": min(timed) if timed else None, "max": max(timed) if timed else None}
    json.dump(out, open(os.path.join(DST, tag + "_pmc_traffic.json"), "w"), indent=1)
    print("dominant:", out["dominant_kernel"], "->", out["hbm_bytes_per_launch"], "B per launch")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01", sys.argv[2] if len(sys.argv) > 2 else "k_solve_islands")
