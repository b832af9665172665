#!/bin/bash
# round 5 session 2: the fused-arithmetic build: GPU suite, bench, island trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5b/gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r5b/gpu_tests.txt
tail -5 gpurun_out/r5b/gpu_tests.txt
timeout 200 python tools/r5/ab_iters.py > gpurun_out/r5b/ab_iters.txt 2>&1; cat gpurun_out/r5b/ab_iters.txt
timeout 300 python tools/island_trace.py > gpurun_out/r5b/island_trace.txt 2>&1; cat gpurun_out/r5b/island_trace.txt
timeout 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r5b/bench.json')); print('ms/step %.4f launch %.2f us' % (d['ms_per_step'], d['roofline']['avg_launch_us']))"
