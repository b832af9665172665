#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/s2/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s2/pytest.log
tail -25 gpurun_out/s2/pytest.log
timeout 300 python tools/island_trace.py > gpurun_out/s2/island_trace.txt 2>&1; cat gpurun_out/s2/island_trace.txt
timeout 300 python tools/island_trace.py 1000 200 20 8 > gpurun_out/s2/island_trace_shard8.txt 2>&1; cat gpurun_out/s2/island_trace_shard8.txt
timeout 300 python tools/island_trace.py 1000 200 0 1 > gpurun_out/s2/island_trace_0it.txt 2>&1; cat gpurun_out/s2/island_trace_0it.txt
