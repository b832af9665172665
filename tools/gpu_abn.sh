#!/bin/bash
# A/B/.. of several builds of the library (phyx_amd/lib_v0.so = current, lib_v1.so ...) in one session
cd $GRAFT_REPO_ROOT
for r in 1 2; do for f in phyx_amd/lib_v*.so; do
  cp $f phyx_amd/libphyx_amd.so
  echo -n "$f: "; timeout 200 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f launch %.2f us' % (d['ms_per_step'], d['roofline']['avg_launch_us']))"
done; done
cp phyx_amd/lib_v0.so phyx_amd/libphyx_amd.so
