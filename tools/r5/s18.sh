#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python tools/r5/group_spread.py > $O/group_spread.txt 2>&1; cat $O/group_spread.txt
timeout 300 python tools/simd_map.py > $O/simd_map.txt 2>&1; tail -22 $O/simd_map.txt
