#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r5v; rm -rf $O; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python tools/fuzz.py 910000 1500 > $O/fuzz_small.txt 2>&1; tail -3 $O/fuzz_small.txt
timeout 700 python tools/fuzz.py 920000 60 --big > $O/fuzz_big.txt 2>&1; tail -3 $O/fuzz_big.txt
timeout 500 python tools/soak.py > $O/soak.txt 2>&1; tail -4 $O/soak.txt
