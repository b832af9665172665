#!/bin/bash
# kernel timeline of one 1M-box world step (cfg 4 size)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/big
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
sed 's/w.set_phase_timing(True)/w.set_phase_timing(False)/' $R/tools/big_world.py > /tmp/big_world_nt.py
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o big -- python /tmp/big_world_nt.py > $O/big.txt 2> $O/big.err
python $R/tools/timeline.py $O/big_kernel_trace.csv k_build_keys -v > $O/big_step_timeline.txt 2>&1
head -4 $O/big_step_timeline.txt; tail -3 $O/big.txt | cut -c1-30
