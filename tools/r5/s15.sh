#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5o; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests/test_broadphase_gpu.py tests/test_world_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt; tail -4 $O/tests.txt
timeout 600 python tools/fuzz.py 73000 80 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
for rep in 1 2; do
timeout 200 python tools/world_quick.py 12 > $O/wq_$rep.txt 2>&1; head -1 $O/wq_$rep.txt
PHX_NO_FUSED_INSERT=1 timeout 200 python tools/world_quick.py 12 > $O/wq_nofi_$rep.txt 2>&1; head -1 $O/wq_nofi_$rep.txt
done
