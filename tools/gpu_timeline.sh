#!/bin/bash
# kernel timeline of one cfg-2 world step (rocprofv3 --kernel-trace of tools/steady.py), printed by tools/timeline.py
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tl
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o world -- python $R/tools/steady.py 12 --no-phase-timing > $O/world_steady.txt 2> $O/world.err
python $R/tools/timeline.py $O/world_kernel_trace.csv k_build_keys -v > $O/world_step_timeline.txt 2>&1
head -30 $O/world_step_timeline.txt
timeout 120 python $R/tools/world_quick.py
