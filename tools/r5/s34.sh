#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
echo small; timeout 200 python tools/r5/ab_iters.py 2>&1 | head -2
echo big; PHX_ISL_SHAPE=big timeout 200 python tools/r5/ab_iters.py 2>&1 | head -2
PHX_ISL_SHAPE=big timeout 300 python tools/island_trace.py 2>&1 | sed -n '1,9p;22,30p'
PHX_ISL_SHAPE=big timeout 200 python tools/class_sizes.py 2>&1 | head -6
