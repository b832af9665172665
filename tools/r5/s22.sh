#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 200 python tools/r5/ab_iters.py > $O/ab_iters.txt 2>&1; cat $O/ab_iters.txt
timeout 300 python tools/island_trace.py > $O/island_trace.txt 2>&1; head -9 $O/island_trace.txt; tail -6 $O/island_trace.txt
PHX_ISL_SHAPE=big timeout 200 python tools/r5/ab_iters.py 2>&1 | head -2
timeout 1500 python -m pytest tests/test_solver_gpu.py -m gpu -q -x --deselect tests/test_solver_gpu.py::test_island_groups_structure > $O/solver_tests.txt 2>&1; echo "pytest rc $?" >> $O/solver_tests.txt; tail -4 $O/solver_tests.txt
