#!/bin/bash
# round 4, session f: N > 1 safety (8 ranks on one GPU, watchdogs, agreement before the all-gather) + the whole GPU suite
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
