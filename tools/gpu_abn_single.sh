#!/bin/bash
# A/B of several builds (phyx_amd/lib_v*.so) on the Single-mode (HBM path) solve of cfg 2
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for f in phyx_amd/lib_v*.so; do
  cp $f phyx_amd/libphyx_amd.so
  echo -n "$f: "; timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['extra']['single_mode']; print('single %.4f ms, launch %.3f us' % (s['ms_per_step'], s['roofline']['avg_launch_us']))"
done; done
