"""One steady World::Update, kernel by kernel: launches, time and measured HBM bytes — from three rocprofv3 runs of the SAME script
(`--kernel-trace`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate passes as MI355X_MICROARCH.md section HBM prescribes).

The step is cut at the marker kernel (the first kernel of an update): the dispatches between its second-to-last and its last
occurrence are the last full steady update of the run.  The script is deterministic, so the three runs dispatch the same kernel
sequence; the cut is made in each file on its own and the sequences are checked against each other.

Counter handling as in tools/pmc_summary.py: FETCH_SIZE / WRITE_SIZE are KiB; the read side is doubled on gfx950.

usage: steady_step_summary.py <kernel_trace.csv> <pmc_fetch_counter_collection.csv> <pmc_write_counter_collection.csv> <out.json> [marker=k_keys_buckets]
"""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# which phase of World::Update a kernel belongs to (ref: World.cpp:19-37); the broadphase rows are what bench.py's cfg 4 roofline sums.
# Helper kernels (scans, mailbox posts, runtime copies / fills) belong to the phase of the named kernel in front of them.
PHASES = (("broadphase", ("k_keys_buckets", "k_bucket_scatter", "k_bucket_sort", "k_build_keys", "k_gather_entries", "k_sweep_rows", "k_sweep_chunks",
                          "k_chunk_bases", "k_emit_pairs", "k_ps_")),
          ("manifolds", ("k_update_manifolds", "k_pack_manifolds", "k_manifold")),
          ("contact_cache", ("k_joints_",)),
          ("schedule", ("k_cc_", "k_joint_components", "k_bin_components", "k_joint_bin_keys_hist", "k_radix_", "k_build_bin", "k_topology_hash")),
          ("solve", ("k_solve_", "k_prestep", "k_pack_refresh", "k_finish_", "k_unpack_bodies")),
          ("integrate", ("k_integrate_",)))
HELPERS = ("k_scan_", "k_post_mail", "__amd_rocclr", "k_upload_words")


def phase_of(name, previous):
    if "RootFlagLoad" in name:                 # (the roots' scan of the schedule build: on the side stream, whatever ran in front of it)
        return "schedule"
    if any(h in name for h in HELPERS):
        return previous
    for ph, keys in PHASES:
        if any(k in name for k in keys):
            return ph
    return "other"


def short(name):
    n = name.split("(")[0]
    return n.replace("phx::", "").replace("void ", "").strip()


def last_step(rows, key, marker):
    rows = sorted(rows, key=key)
    at = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(at) < 2:
        raise SystemExit("marker %r occurs %d times: need two" % (marker, len(at)))
    return rows[at[-2]:at[-1]]


def main():
    trace, fetch, write, out_path = sys.argv[1:5]
    marker = sys.argv[5] if len(sys.argv) > 5 else "k_keys_buckets"
    t = last_step(list(csv.DictReader(open(trace))), lambda r: int(r["Start_Timestamp"]), marker)
    f = last_step(list(csv.DictReader(open(fetch))), lambda r: int(r["Dispatch_Id"]), marker)
    w = last_step(list(csv.DictReader(open(write))), lambda r: int(r["Dispatch_Id"]), marker)
    names = [short(r["Kernel_Name"]) for r in t]
    # (the step runs on two streams since round 5 — the components' labels on the solver's side stream beside RefreshContactJoints — so
    #  the three passes interleave the two streams' dispatches differently: the k-th dispatch of a kernel in one pass is the k-th of that
    #  kernel in another; the passes must agree on how often every kernel runs)
    def by_name(rows):
        d = collections.defaultdict(list)
        for r in rows:
            d[short(r["Kernel_Name"])].append(r)
        return d
    fn, wn = by_name(f), by_name(w)
    for other, what in ((fn, "FETCH_SIZE"), (wn, "WRITE_SIZE")):
        if {k: len(v) for k, v in other.items()} != dict(collections.Counter(names)):
            raise SystemExit("the %s pass dispatched other kernels than the kernel trace (%d vs %d dispatches): not the same step" % (what, sum(len(v) for v in other.values()), len(names)))
    seen = collections.Counter()
    f2, w2 = [], []
    for n in names:
        f2.append(fn[n][seen[n]]); w2.append(wn[n][seen[n]]); seen[n] += 1
    f, w = f2, w2
    agg = collections.OrderedDict()
    phase = "broadphase"
    for rt, rf, rw in zip(t, f, w):
        k = short(rt["Kernel_Name"])
        phase = phase_of(k, phase)
        a = agg.setdefault((k, phase), {"kernel": k, "phase": phase, "launches": 0, "us": 0.0, "fetch_bytes_raw": 0.0, "write_bytes": 0.0})
        a["launches"] += 1
        a["us"] += (int(rt["End_Timestamp"]) - int(rt["Start_Timestamp"])) / 1e3
        a["fetch_bytes_raw"] += float(rf["Counter_Value"]) * 1024
        a["write_bytes"] += float(rw["Counter_Value"]) * 1024
    kernels = []
    for a in agg.values():
        a["hbm_bytes"] = 2 * a["fetch_bytes_raw"] + a["write_bytes"]
        a["GBps"] = a["hbm_bytes"] / (a["us"] * 1e-6) / 1e9 if a["us"] > 0 else None
        a["us"] = round(a["us"], 2)
        kernels.append(a)
    t0 = int(t[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in t)
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = "unrecorded"
    phases = collections.OrderedDict()
    for a in kernels:
        p = phases.setdefault(a["phase"], {"launches": 0, "us": 0.0, "hbm_bytes": 0.0})
        p["launches"] += a["launches"]; p["us"] += a["us"]; p["hbm_bytes"] += a["hbm_bytes"]
    for p in phases.values():
        p["us"] = round(p["us"], 2)
        p["frac_of_8TBps"] = p["hbm_bytes"] / (p["us"] * 1e-6) / 8e12 if p["us"] > 0 else None
    out = {"what": "one steady World::Update (the dispatches between the last two %s): per kernel its launches in that update, the sum of their "
                   "kernel times (rocprofv3 --kernel-trace) and of their HBM bytes (rocprofv3 --pmc FETCH_SIZE x 2 + --pmc WRITE_SIZE, separate passes "
                   "of the same deterministic script)" % marker,
           "commit": commit, "marker": marker, "kernels_in_step": len(t), "span_us": round((t1 - t0) / 1e3, 1),
           "busy_us": round(sum(a["us"] for a in kernels), 1), "phases": phases, "kernels": kernels}
    json.dump(out, open(out_path, "w"), indent=1)
    print("%d kernels, span %.1f us, busy %.1f us" % (len(t), out["span_us"], out["busy_us"]))
    for ph, p in phases.items():
        print("  %-14s %3d launches %8.1f us %9.1f MB  %.3f of 8 TB/s" % (ph, p["launches"], p["us"], p["hbm_bytes"] / 1e6, p["frac_of_8TBps"] or 0))
    for a in kernels:
        print("    %-44s x%-3d %8.1f us %9.2f MB" % (a["kernel"][:44], a["launches"], a["us"], a["hbm_bytes"] / 1e6))


if __name__ == "__main__":
    main()
