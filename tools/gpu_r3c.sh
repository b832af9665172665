#!/bin/bash
# settled-world regime: phase times at steps ~56-60, and a kernel timeline of one settled step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
export PYTHONUNBUFFERED=1
timeout 600 python tools/steady.py 60 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3c -o settled -- python $GRAFT_REPO_ROOT/tools/steady.py 52 --no-phase-timing > $GRAFT_REPO_ROOT/gpurun_out/r3c/steady.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py gpurun_out/r3c/settled_kernel_trace.csv k_build_keys -v > gpurun_out/r3c/settled_timeline.txt 2>&1
head -12 gpurun_out/r3c/settled_timeline.txt
awk '{print $3, $4}' gpurun_out/r3c/settled_timeline.txt | sort | uniq -c | sort -rn | head -30
