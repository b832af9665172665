#!/bin/bash
# round 4, session p: kernel-by-kernel timeline of a settled-world step (gaps = host round trips)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o settled -- python $R/tools/steady.py 62 --no-phase-timing > $O/steady.txt 2> $O/err.txt
python $R/tools/timeline.py $O/settled_kernel_trace.csv k_keys_buckets -v > $O/settled_timeline.txt 2>&1
head -30 $O/settled_timeline.txt
