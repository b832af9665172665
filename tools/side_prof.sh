#!/bin/bash
# rocprofv3 passes of the side configurations (tools/prof_cfg.py): cfg 5 (500k boxes, 50 iterations) — kernel stats + the two PMC passes, named
# as tools/pmc_summary.py expects them under gpurun_out/prof (run it afterwards: profiles/<tag>_pmc_traffic_cfg5.json, <tag>_cfg5_kernel_stats.csv)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in cfg5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ${c}_trace -- python $R/tools/prof_cfg.py $c > $O/${c}_trace.txt 2> $O/${c}_trace.err
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o ${c}_pmc_fetch -- python $R/tools/prof_cfg.py $c > $O/${c}_fetch.txt 2> $O/${c}_fetch.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o ${c}_pmc_write -- python $R/tools/prof_cfg.py $c > $O/${c}_write.txt 2> $O/${c}_write.err
done
ls $O | grep cfg5
