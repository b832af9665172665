#!/bin/bash
# quick GPU check: solver + world parity tests, island trace, short bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_solver_gpu.py tests/test_world_gpu.py -m gpu -q -x > gpurun_out/q/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/q/pytest.log
tail -8 gpurun_out/q/pytest.log
timeout 300 python tools/island_trace.py > gpurun_out/q/island_trace.txt 2>&1; cat gpurun_out/q/island_trace.txt
timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/q/bench.json 2> gpurun_out/q/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/q/bench.json'))
print("ms/step",d["ms_per_step"],"value %.3g"%d["value"],"launch us",d["roofline"]["avg_launch_us"])
PY
