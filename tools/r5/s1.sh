#!/bin/bash
# round 5 session 1: A/B of lib variants (v0 current, v1 = islands.hip under -ffp-contract=fast, timing only) + sweep-mix times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
for f in phyx_amd/lib_v*.so; do
  cp $f phyx_amd/libphyx_amd.so
  echo "== $f" >> gpurun_out/r5a/ab.txt
  timeout 200 python tools/r5/ab_iters.py >> gpurun_out/r5a/ab.txt 2>&1
  for r in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f launch %.2f us' % (d['ms_per_step'], d['roofline']['avg_launch_us']))" >> gpurun_out/r5a/ab.txt
  done
done
cp phyx_amd/lib_v0.so phyx_amd/libphyx_amd.so
cat gpurun_out/r5a/ab.txt
