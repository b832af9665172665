#!/bin/bash
# round 4, session l: quick parity (solver + world suites, -x) and the bench's side measurements
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_solver_gpu.py tests/test_world_gpu.py -m gpu -x -q > gpurun_out/r4l/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/r4l/pytest.log | tail -8
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r4l/bench.json 2> gpurun_out/r4l/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4l/bench.json'))
print("ms/step",d["ms_per_step"],"value %.3g"%d["value"],"launch us",d["roofline"]["avg_launch_us"])
e=d["extra"]["other_configs"]
print("live",d["live_topology"]["ms_per_step"],"single",d["single_mode"]["ms_per_step"],"world",e["cfg2_world_step"]["ms_per_step"],
      "settled",e["settled_world_step"]["ms_per_step"],"cfg4 step",e["cfg4_broadphase_1M"]["world_step_ms"],"cfg5",e["cfg5_500k_tall_50it_fp32"]["ms_per_step"], "fp16", e["cfg5_500k_tall_50it_fp16_body_state"]["ms_per_step"])
PY
