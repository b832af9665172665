#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
export PYTHONUNBUFFERED=1
PHX_ISL_SHAPE=big timeout 200 python tools/r5/ab_iters.py > $O/ab_iters_big.txt 2>&1; cat $O/ab_iters_big.txt
PHX_ISL_SHAPE=big timeout 300 python tools/island_trace.py > $O/island_trace_big.txt 2>&1; cat $O/island_trace_big.txt | head -12; tail -6 $O/island_trace_big.txt
PHX_ISL_SHAPE=big timeout 300 python -m pytest tests/test_solver_gpu.py -m gpu -q -x -k "full_size_200k or device_schedule_builder" 2>&1 | tail -5
