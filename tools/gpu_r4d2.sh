#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_broadphase_gpu.py -m gpu -x -q -k "two_level or persistent or first_update" 2>&1 | tail -4
for v in lsd split lsd split; do
  if [ $v = lsd ]; then export PHX_NO_SPLIT_SORT=1; else unset PHX_NO_SPLIT_SORT; fi
  echo $v $(timeout 300 python tools/world_quick.py 2>&1 | tail -1)
done
unset PHX_NO_SPLIT_SORT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4d -o world -- python $GRAFT_REPO_ROOT/tools/steady.py 12 --no-phase-timing > $GRAFT_REPO_ROOT/gpurun_out/r4d/world_steady.txt 2> $GRAFT_REPO_ROOT/gpurun_out/r4d/world.err
python $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/gpurun_out/r4d/world_kernel_trace.csv k_keys_buckets -v > $GRAFT_REPO_ROOT/gpurun_out/r4d/world_step_timeline.txt 2>&1
grep -E "k_keys_buckets|k_bucket_s|step span" $GRAFT_REPO_ROOT/gpurun_out/r4d/world_step_timeline.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4d -o cfg4 -- python $GRAFT_REPO_ROOT/tools/prof_cfg.py cfg4 > $GRAFT_REPO_ROOT/gpurun_out/r4d/cfg4.txt 2> $GRAFT_REPO_ROOT/gpurun_out/r4d/cfg4.err
python $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/gpurun_out/r4d/cfg4_kernel_trace.csv k_keys_buckets -v > $GRAFT_REPO_ROOT/gpurun_out/r4d/cfg4_step_timeline.txt 2>&1
grep -E "k_keys_buckets|k_bucket_s|step span" $GRAFT_REPO_ROOT/gpurun_out/r4d/cfg4_step_timeline.txt
