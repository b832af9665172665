#!/bin/bash
# round 4, session q: island kernel write traffic (PMC WRITE_SIZE / FETCH_SIZE) and launch time, quick
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; rm -rf $O; mkdir -p $O
ARGS="--steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary"
timeout 600 python $R/bench.py $ARGS > $O/bench.json 2> $O/bench.err
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O -o pmc_$c -- python $R/bench.py $ARGS > /dev/null 2> $O/$c.err
done
python - <<'PY'
import csv, json, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4q"
d=json.load(open(O+"/bench.json")); print("ms/step",round(d["ms_per_step"],4),"launch us",round(d["roofline"]["avg_launch_us"],2))
for c in ("WRITE_SIZE","FETCH_SIZE"):
    acc=collections.Counter(); n=collections.Counter()
    for r in csv.DictReader(open(O+"/pmc_%s_counter_collection.csv"%c)):
        if "k_solve_islands" in r["Kernel_Name"] and r["Grid_Size"] in ("259840","256000"):
            acc[r["Grid_Size"]]+=float(r["Counter_Value"]); n[r["Grid_Size"]]+=1
    for g in acc: print(c, g, n[g], "launches", round(acc[g]/n[g]/1e6,2), "M units per launch")
PY
