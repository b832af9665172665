#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python tools/island_trace.py > $O/island_trace.txt 2>&1; cat $O/island_trace.txt | head -60
timeout 200 python tools/r5/ab_iters.py > $O/ab_iters.txt 2>&1; cat $O/ab_iters.txt
timeout 200 python tools/class_sizes.py > $O/class_sizes.txt 2>&1; head -40 $O/class_sizes.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('ms/step %.4f launch %.2f us live %.4f world %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_us'], d['live_topology_ms_per_step'], d['world_step_ms_per_step']))"; tail -3 $O/bench.err
