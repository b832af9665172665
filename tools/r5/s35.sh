#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_prof.sh > gpurun_out/prof_run.log 2>&1; tail -5 gpurun_out/prof_run.log
bash tools/side_prof.sh > gpurun_out/side_run.log 2>&1; tail -5 gpurun_out/side_run.log
bash tools/sq_pass.sh > gpurun_out/sq_run.log 2>&1; tail -12 gpurun_out/sq_run.log
bash tools/r5/steady_prof.sh r5p > gpurun_out/steady_run.log 2>&1; grep -v "^    " gpurun_out/steady_run.log | head -24
