#!/bin/bash
# one steady World::Update kernel by kernel (time + HBM bytes): cfg 4 (1M boxes) and the running cfg 2 world
# usage: steady_prof.sh <outdir under gpurun_out> ; leaves <outdir>/{cfg4,cfg2w}_steady_step.json
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in cfg4 cfg2w; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o ${c}_trace -- python $R/tools/prof_cfg.py $c > $O/${c}_trace.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o ${c}_fetch -- python $R/tools/prof_cfg.py $c > $O/${c}_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o ${c}_write -- python $R/tools/prof_cfg.py $c > $O/${c}_write.log 2>&1
  python $R/tools/steady_step_summary.py $O/${c}_trace_kernel_trace.csv $O/${c}_fetch_counter_collection.csv $O/${c}_write_counter_collection.csv $O/${c}_steady_step.json
  python $R/tools/timeline.py $O/${c}_trace_kernel_trace.csv k_keys_buckets -v > $O/${c}_step_timeline.txt 2>&1
  rm -f $O/${c}_*_agent_info.csv
done
ls -la $O | head -30
