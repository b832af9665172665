#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
( timeout 900 python tools/fuzz.py 300000 2500 > gpurun_out/fuzz/small2.log 2>&1; echo "small rc=$?" >> gpurun_out/fuzz/small2.log ) &
( timeout 900 python tools/fuzz.py 400000 60 --big > gpurun_out/fuzz/big2.log 2>&1; echo "big rc=$?" >> gpurun_out/fuzz/big2.log ) &
for i in 1 2 3; do timeout 100 python tools/world_quick.py; done
wait
tail -2 gpurun_out/fuzz/small2.log; tail -2 gpurun_out/fuzz/big2.log
