#!/bin/bash
# quick GPU check: solver + world parity tests, short bench (island launch, Single mode)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_solver_gpu.py tests/test_world_gpu.py -m gpu -q -x 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/q/bench.json 2> gpurun_out/q/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/q/bench.json'))
e=d["extra"]
print("ms/step",round(d["ms_per_step"],4),"value %.4g"%d["value"],"launch us",round(d["roofline"]["avg_launch_us"],2), "single", round(d["single_mode"]["ms_per_step"],3), "live", round(d["live_topology"]["ms_per_step"],3))
PY
