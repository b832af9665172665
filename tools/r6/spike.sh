#!/bin/bash
# kernel timeline of step N-1 of the 200k world (series.py N): usage spike.sh N...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/spike; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o s$n -- python $R/tools/r6/series.py $n > $O/s$n.log 2>&1
python $R/tools/timeline.py $O/s${n}_kernel_trace.csv k_keys_buckets -v > $O/s${n}_timeline.txt 2>&1
head -2 $O/s${n}_timeline.txt
done
rm -f $O/*_kernel_trace.csv $O/*agent_info.csv
