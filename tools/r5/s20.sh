#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python tools/island_trace.py > $O/island_trace.txt 2>&1; cat $O/island_trace.txt | head -40
