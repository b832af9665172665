#!/bin/bash
# settled world step with and without the partitioned-component path
mkdir -p gpurun_out/parts
for np in 0 1; do
  echo "== PHX_NO_PARTS=$np"
  PHX_NO_PARTS=$np timeout 300 python tools/steady.py 64 --no-phase-timing 2>&1 | tail -8
done > gpurun_out/parts/steady.txt 2>&1
PHX_TRACE_SCHEDULE=1 timeout 300 python tools/steady.py 60 --no-phase-timing 2>&1 | grep -v "^\[schedule" | tail -3 >> gpurun_out/parts/steady.txt
PHX_TRACE_SCHEDULE=1 timeout 300 python tools/steady.py 60 --no-phase-timing 2>&1 | grep "^\[schedule" | tail -40 > gpurun_out/parts/schedule_trace.txt
