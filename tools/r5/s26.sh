#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
echo twins; timeout 200 python tools/r5/ab_iters.py 2>&1 | head -2
echo no twins; PHX_NO_TWINS=1 timeout 200 python tools/r5/ab_iters.py 2>&1 | head -2
PHX_NO_TWINS=1 timeout 200 python tools/class_sizes.py | head -4
PHX_NO_TWINS=1 timeout 300 python tools/world_quick.py 30
timeout 300 python tools/world_quick.py 30
