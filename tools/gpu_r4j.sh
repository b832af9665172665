#!/bin/bash
# round 4, session j: connected components by lock-free union-find — full GPU suite, then the bench's side measurements
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r4j/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4j/pytest.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r4j/pytest.log | tail -15
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4j/bench.json'))
print("ms/step",d["ms_per_step"],"value %.3g"%d["value"],"launch us",d["roofline"]["avg_launch_us"])
e=d["extra"]["other_configs"]
print("live",d["live_topology"]["ms_per_step"],"single",d["single_mode"]["ms_per_step"],"world",e["cfg2_world_step"]["ms_per_step"],
      "settled",e["settled_world_step"]["ms_per_step"],"cfg4 step",e["cfg4_broadphase_1M"]["world_step_ms"],"cfg5",e["cfg5_500k_tall_50it_fp32"]["ms_per_step"])
PY
