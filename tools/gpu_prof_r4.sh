#!/bin/bash
# round 4 rocprofv3 evidence for profiles/: bench.py (cfg 2: kernel trace, main-only trace, PMC FETCH / WRITE in separate passes),
# the world step (timeline + marker stats), the cfg 4 and cfg 5 workloads (tools/prof_cfg.py) and the settled world.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --repeats 3 --no-cpu-baseline"
timeout 1200 python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o trace -- python $R/bench.py $ARGS > $O/bench_trace.json 2> $O/trace.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o main -- python $R/bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_main.json 2> $O/main.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o pmc_fetch -- python $R/bench.py $ARGS > $O/bench_pmc_fetch.json 2> $O/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o pmc_write -- python $R/bench.py $ARGS > $O/bench_pmc_write.json 2> $O/pmc_write.err
# (two passes: with the markers traced, every roctx range costs the host microseconds and the kernel timeline grows gaps that a plain run does not have)
timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O -o worldm -- python $R/tools/steady.py 12 --no-phase-timing > $O/world_steady_markers.txt 2> $O/worldm.err
cp $O/worldm_marker_api_stats.csv $O/world_marker_api_stats.csv 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o world -- python $R/tools/steady.py 12 --no-phase-timing > $O/world_steady.txt 2> $O/world.err
python $R/tools/timeline.py $O/world_kernel_trace.csv k_keys_buckets -v > $O/world_step_timeline.txt 2>&1
for c in cfg4 cfg5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ${c}_trace -- python $R/tools/prof_cfg.py $c > $O/${c}_trace.txt 2> $O/${c}_trace.err
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o ${c}_pmc_fetch -- python $R/tools/prof_cfg.py $c > $O/${c}_fetch.txt 2> $O/${c}_fetch.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o ${c}_pmc_write -- python $R/tools/prof_cfg.py $c > $O/${c}_write.txt 2> $O/${c}_write.err
done
python $R/tools/timeline.py $O/cfg4_trace_kernel_trace.csv k_keys_buckets -v > $O/cfg4_step_timeline.txt 2>&1
head -3 $O/world_step_timeline.txt
head -3 $O/cfg4_step_timeline.txt
# the settled world (one merged island, the partitioned-component path): kernel statistics + a per-kernel breakdown of step 62 + PMC
$R/tools/gpu_parts_prof.sh 62 pmc > /dev/null 2>&1
cp $R/gpurun_out/parts/last_step.txt $O/settled_last_step.txt; cp $R/gpurun_out/parts/kernel_stats.csv $O/settled_kernel_stats.csv
head -5 $O/settled_last_step.txt
python - <<PY
import json
d=json.load(open("$O/bench_plain.json"))
print("ms/step",round(d["ms_per_step"],4),"value %.4g"%d["value"],"launch us",round(d["roofline"]["avg_launch_us"],2),"single",round(d["single_mode"]["ms_per_step"],3),"live",round(d["live_topology"]["ms_per_step"],3))
oc=d["extra"]["other_configs"]
print({k:(round(v.get("ms_per_step",0),3) if isinstance(v,dict) and "ms_per_step" in v else None) for k,v in oc.items()})
print("cpu",d.get("cpu_baseline",{}).get("value"),d.get("cpu_baseline",{}).get("cores"))
PY
