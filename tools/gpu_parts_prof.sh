#!/bin/bash
# kernel-level view of the settled world's solve: rocprofv3 --kernel-trace --stats of tools/steady.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/parts
rm -rf /tmp/pp && mkdir -p /tmp/pp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pp -o parts --output-format csv -- python $R/tools/steady.py ${1:-60} --no-phase-timing > $R/gpurun_out/parts/prof_stdout.txt 2>&1
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/parts/kernel_stats.csv
head -25 "$f" | cut -c1-200
tail -3 $R/gpurun_out/parts/prof_stdout.txt | cut -c1-30,330-
t=$(find /tmp/pp -name "*kernel_trace.csv" | head -1)
python - "$t" > $R/gpurun_out/parts/last_step.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last world step starts at the last k_build_keys launch
idx = max(i for i, r in enumerate(rows) if "k_build_keys" in r["Kernel_Name"] or "k_keys_buckets" in r["Kernel_Name"])
step = rows[idx:]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
agg = collections.OrderedDict()
for r in step:
    n = r["Kernel_Name"].split("(")[0].replace("void phx::", "").replace("phx::", "")
    a = agg.setdefault(n, [0, 0.0, (int(r["Start_Timestamp"]) - t0) / 1e3])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
busy = sum(a[1] for a in agg.values())
print("last step: span %.1f us, %d kernels, busy %.1f us" % ((t1 - t0) / 1e3, len(step), busy))
for n, a in agg.items():
    print("%8.1f us first  %4d x  %8.1f us total  %s" % (a[2], a[0], a[1], n))
PY
cat $R/gpurun_out/parts/last_step.txt | head -80
# PMC traffic of the same run (separate passes, --pmc only): gpurun_out/prof/settled_pmc_{fetch,write}_counter_collection.csv for
# tools/pmc_summary.py (side_config "settled")
if [ "${2:-}" = "pmc" ]; then
  mkdir -p $R/gpurun_out/prof
  for c in fetch write; do
    C=$([ $c = fetch ] && echo FETCH_SIZE || echo WRITE_SIZE)
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/prof -o settled_pmc_$c -- python $R/tools/steady.py ${1:-60} --no-phase-timing > /dev/null 2>&1
  done
  cp "$f" $R/gpurun_out/prof/settled_trace_kernel_stats.csv
fi
