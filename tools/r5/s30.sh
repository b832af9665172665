#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
echo base; timeout 200 python tools/r5/ab_iters.py 2>&1 | head -2
cp phyx_amd/libphyx_amd.so /tmp/keep.so; cp _slp/libphyx_amd.so phyx_amd/libphyx_amd.so
echo slp; timeout 200 python tools/r5/ab_iters.py 2>&1 | head -2
timeout 600 python -m pytest tests/test_solver_gpu.py -m gpu -q -x 2>&1 | tail -2
cp /tmp/keep.so phyx_amd/libphyx_amd.so
echo base; timeout 200 python tools/r5/ab_iters.py 2>&1 | head -2
