#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 200 python tools/class_sizes.py > $O/class_sizes.txt 2>&1; head -16 $O/class_sizes.txt
timeout 200 python tools/r5/ab_iters.py > $O/ab_iters.txt 2>&1; cat $O/ab_iters.txt
timeout 1500 python -m pytest tests/test_solver_gpu.py -m gpu -q -x --deselect tests/test_solver_gpu.py::test_island_groups_structure > $O/solver_tests.txt 2>&1; echo "pytest rc $?" >> $O/solver_tests.txt; tail -40 $O/solver_tests.txt
