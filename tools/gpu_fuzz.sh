#!/bin/bash
# differential fuzz + soak of the device World against the oracle World (every byte, every step)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
( timeout 1500 python tools/fuzz.py 100000 6000 > gpurun_out/fuzz/small.log 2>&1; echo "small rc=$?" >> gpurun_out/fuzz/small.log ) &
( timeout 1500 python tools/fuzz.py 200000 150 --big > gpurun_out/fuzz/big.log 2>&1; echo "big rc=$?" >> gpurun_out/fuzz/big.log ) &
( timeout 1500 python tools/soak.py > gpurun_out/fuzz/soak.log 2>&1; echo "soak rc=$?" >> gpurun_out/fuzz/soak.log ) &
wait
tail -3 gpurun_out/fuzz/small.log; tail -3 gpurun_out/fuzz/big.log; tail -4 gpurun_out/fuzz/soak.log
