"""SQ counter passes of tools/sq_pass.sh -> one JSON: per pass the island kernel's full-grid launches (most frequent grid), mean launch time
(from the pass's own kernel trace) and the mean of every counter per launch (sums over the chip); a few derived figures.
usage: sq_summary.py <dir with sq{1,2,3}_counter_collection.csv / sq{1,2,3}_kernel_trace.csv> <out.json>"""
import collections, csv, json, os, sys

src, out_path = sys.argv[1], sys.argv[2]
KERNEL = "k_solve_islands<256"
passes = {}
for tag in ("sq1", "sq2", "sq3"):
    cc = os.path.join(src, tag + "_counter_collection.csv")
    if not os.path.exists(cc):
        continue
    rows = [r for r in csv.DictReader(open(cc)) if KERNEL in r["Kernel_Name"]]
    if not rows:
        continue
    grid = collections.Counter(r["Grid_Size"] for r in rows).most_common(1)[0][0]
    rows = [r for r in rows if r["Grid_Size"] == grid]
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    launches = max(len(v) for v in per.values())
    us = None
    kt = os.path.join(src, tag + "_kernel_trace.csv")
    if os.path.exists(kt):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if KERNEL in r["Kernel_Name"] and r.get("Grid_Size", grid) == grid]
        if d:
            us = sum(d) / len(d)
    passes[tag] = {"launches": launches, "grid_size": grid, "mean_launch_us": us, "counters": {k: sum(v) / len(v) for k, v in sorted(per.items())}}
c = {}
for p in passes.values():
    c.update(p["counters"])
derived = {}
waves = c.get("SQ_WAVES")
if waves and c.get("SQ_WAVE_CYCLES"):
    derived["waves_per_launch"] = waves
    derived["wave_cycles_per_wave"] = c["SQ_WAVE_CYCLES"] / waves
if c.get("SQ_INSTS_VALU") and waves:
    groups = waves / 4.0
    derived["valu_instructions_per_group"] = c["SQ_INSTS_VALU"] / groups
    derived["valu_instructions_per_group_and_class_step_(84_steps)"] = c["SQ_INSTS_VALU"] / groups / 84.0
if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_INSTS_VALU"):
    derived["cycles_per_valu_instruction_(ACTIVE_INST_VALU/INSTS_VALU*4)"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
if c.get("SQ_WAVE_CYCLES"):
    share = {}
    for name, key in (("waiting (s_waitcnt / s_barrier: SQ_WAIT_ANY)", "SQ_WAIT_ANY"), ("issue stalls (SQ_WAIT_INST_ANY)", "SQ_WAIT_INST_ANY"), ("issuing (SQ_ACTIVE_INST_ANY)", "SQ_ACTIVE_INST_ANY")):
        if c.get(key) is not None:
            share[name] = c[key] / c["SQ_WAVE_CYCLES"]
    derived["share_of_wave_cycles"] = share
us = next((p["mean_launch_us"] for p in passes.values() if p["mean_launch_us"]), None)
if us and c.get("SQ_INSTS_VALU"):
    # a SIMD issues one wave64 fp32 operation per 4 cycles (16 lanes): 1024 SIMDs at 2.4 GHz
    derived["valu_busy_share_of_simd_time_at_4_cycles_per_operation"] = c["SQ_INSTS_VALU"] * 4.0 / (1024 * us * 1e-6 * 2.4e9)
json.dump({"what": "rocprofv3 --pmc SQ passes of `bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary` (cfg 2), kernel k_solve_islands<256,768,false,false>, "
                   "full-grid launches; counters are sums over the chip; SQ_*CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (4 shader cycles)",
           "passes": passes, "derived": derived}, open(out_path, "w"), indent=1)
print(json.dumps(derived, indent=1))
