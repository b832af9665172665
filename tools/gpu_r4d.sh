#!/bin/bash
# round 4, session d: the two-level broadphase sort — parity (broadphase + world lockstep), A/B of the world step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_broadphase_gpu.py tests/test_world_gpu.py -m gpu -x -q 2>&1 | tail -8
for v in lsd split lsd split; do
  if [ $v = lsd ]; then export PHX_NO_SPLIT_SORT=1; else unset PHX_NO_SPLIT_SORT; fi
  echo $v $(timeout 300 python tools/world_quick.py 2>&1 | tail -1)
done
unset PHX_NO_SPLIT_SORT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4d -o world -- python $GRAFT_REPO_ROOT/tools/steady.py 12 --no-phase-timing > $GRAFT_REPO_ROOT/gpurun_out/r4d/world_steady.txt 2> $GRAFT_REPO_ROOT/gpurun_out/r4d/world.err
python $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/gpurun_out/r4d/world_kernel_trace.csv k_keys_buckets -v > $GRAFT_REPO_ROOT/gpurun_out/r4d/world_step_timeline.txt 2>&1
head -30 $GRAFT_REPO_ROOT/gpurun_out/r4d/world_step_timeline.txt
