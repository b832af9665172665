#!/bin/bash
# A/B of two builds of the library (phyx_amd/lib_a.so, lib_b.so) in one session, alternating
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in a b; do
  cp phyx_amd/lib_$v.so phyx_amd/libphyx_amd.so
  echo -n "$v: "; timeout 200 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f launch %.2f us' % (d['ms_per_step'], d['roofline']['avg_launch_us']))"
done; done
