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
