#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_solver_gpu.py -m gpu -x -q > $O/solver_tests.txt 2>&1; echo "pytest rc $?" >> $O/solver_tests.txt; tail -25 $O/solver_tests.txt
timeout 200 python tools/r5/ab_iters.py > $O/ab_iters.txt 2>&1; cat $O/ab_iters.txt
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('ms/step %.4f launch %.2f us' % (d['ms_per_step'], d['roofline']['avg_launch_us'])); print(d['config']['colours'])"; tail -3 $O/bench.err
