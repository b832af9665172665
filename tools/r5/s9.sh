#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5i; mkdir -p $O
export PYTHONUNBUFFERED=1
for sc in stack merge falling tilted; do echo -n "$sc inc: "; timeout 300 python tools/incremental_twin.py $sc 45 2>&1 | tail -1; echo -n "$sc full: "; PHX_NO_INCREMENTAL=1 timeout 300 python tools/incremental_twin.py $sc 45 2>&1 | tail -1; done 2>&1 | tee $O/twin.txt
timeout 200 python tools/world_quick.py > $O/world_quick.txt 2>&1; tail -1 $O/world_quick.txt
PHX_NO_INCREMENTAL=1 timeout 200 python tools/world_quick.py > $O/world_quick_noinc.txt 2>&1; tail -1 $O/world_quick_noinc.txt
PHX_NO_MAIL_CARRIER=1 timeout 200 python tools/world_quick.py > $O/world_quick_nocarrier.txt 2>&1; tail -1 $O/world_quick_nocarrier.txt
timeout 1800 python -m pytest tests/test_world_gpu.py tests/test_broadphase_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt; tail -5 $O/tests.txt
timeout 600 python tools/fuzz.py 72000 80 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
bash tools/r5/steady_prof.sh r5i > $O/steady_prof.log 2>&1; grep -v "^    " $O/steady_prof.log | head -24
