#!/bin/bash
# kernel statistics of a long-settled world (merged-island regime): rocprofv3 --kernel-trace --stats of tools/steady.py
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sp
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o steady -- python $R/tools/steady.py 48 --no-phase-timing > $O/steady.txt 2> $O/steady.err
python $R/tools/timeline.py $O/steady_kernel_trace.csv k_build_keys -v > $O/steady_last_step_timeline.txt 2>&1
head -40 $O/steady_last_step_timeline.txt
tail -3 $O/steady.txt | cut -c1-20
