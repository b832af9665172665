#!/bin/bash
# timing experiments: variants of the library built with -DPHX_EXP=n (not bit-exact, timing only)
cd $GRAFT_REPO_ROOT
for n in 0 1 2 3 4 5; do
  cp phyx_amd/libphyx_exp$n.so phyx_amd/libphyx_amd.so
  echo -n "EXP $n: "; timeout 200 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f launch %.2f us' % (d['ms_per_step'], d['roofline']['avg_launch_us']))"
done
cp phyx_amd/libphyx_exp0.so phyx_amd/libphyx_amd.so
