#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2; do
timeout 200 python tools/world_quick.py 36 -v > $O/wq_$rep.txt 2>&1; tail -3 $O/wq_$rep.txt
PHX_NO_INCREMENTAL=1 timeout 200 python tools/world_quick.py 36 -v > $O/wq_noinc_$rep.txt 2>&1; tail -3 $O/wq_noinc_$rep.txt
PHX_NO_MAIL_CARRIER=1 timeout 200 python tools/world_quick.py 36 > $O/wq_nocarrier_$rep.txt 2>&1; tail -2 $O/wq_nocarrier_$rep.txt
done
timeout 1200 python -m pytest tests/test_solver_gpu.py -m gpu -x -q 2>&1 | tail -3
