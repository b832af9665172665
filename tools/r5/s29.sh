#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5t; rm -rf $O; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 200 python tools/r5/ab_iters.py 2>&1 | head -2
timeout 300 python tools/island_trace.py > $O/island_trace.txt 2>&1; head -9 $O/island_trace.txt; tail -6 $O/island_trace.txt
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -5 $O/gpu_tests.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('ms/step %.4f launch %.2f us live %.4f world %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_us'], d['live_topology_ms_per_step'], d['world_step_ms_per_step']))"; tail -3 $O/bench.err
