#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_world_gpu.py tests/test_broadphase_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt; tail -5 $O/tests.txt
timeout 600 python tools/fuzz.py 71000 60 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
bash tools/r5/steady_prof.sh r5h > $O/steady_prof.log 2>&1; grep -v "^    " $O/steady_prof.log | head -30; grep "k_keys_buckets\|k_bucket_\|k_update_manifolds\|k_joints_match\|k_post_mail" $O/steady_prof.log
timeout 200 python tools/world_quick.py > $O/world_quick.txt 2>&1; tail -3 $O/world_quick.txt
PHX_NO_MAIL_CARRIER=1 timeout 200 python tools/world_quick.py > $O/world_quick_nocarrier.txt 2>&1; tail -1 $O/world_quick_nocarrier.txt
