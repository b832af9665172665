#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_solver_gpu.py -m gpu -x -q > $O/solver_tests.txt 2>&1; echo "pytest rc $?" >> $O/solver_tests.txt; tail -3 $O/solver_tests.txt
timeout 200 python tools/r5/ab_iters.py > $O/ab_iters.txt 2>&1; cat $O/ab_iters.txt
for r in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('ms/step %.4f launch %.2f us' % (d['ms_per_step'], d['roofline']['avg_launch_us']))"; done
timeout 300 python tools/island_trace.py > $O/island_trace.txt 2>&1; tail -8 $O/island_trace.txt
