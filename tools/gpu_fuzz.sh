#!/bin/bash
# differential fuzz + soak of the device World against the oracle World (every byte, every step); three processes share the GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
F=${1:-500000}; N=${2:-6000}; B=${3:-150}
( timeout 2400 python tools/fuzz.py $F $N > gpurun_out/fuzz/small.log 2>&1; echo "small rc=$?" >> gpurun_out/fuzz/small.log ) &
( timeout 2400 python tools/fuzz.py $((F + 100000)) $B --big > gpurun_out/fuzz/big.log 2>&1; echo "big rc=$?" >> gpurun_out/fuzz/big.log ) &
( timeout 2400 python tools/soak.py > gpurun_out/fuzz/soak.log 2>&1; echo "soak rc=$?" >> gpurun_out/fuzz/soak.log ) &
wait
tail -2 gpurun_out/fuzz/small.log; tail -2 gpurun_out/fuzz/big.log; tail -2 gpurun_out/fuzz/soak.log
