#!/bin/bash
# whole GPU parity suite + island trace + default bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/f/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/f/pytest.log
grep -E "passed|failed|FAILED|Error" gpurun_out/f/pytest.log | tail -15
timeout 300 python tools/island_trace.py 2>&1 | grep -v XCC > gpurun_out/f/island_trace.txt; cat gpurun_out/f/island_trace.txt
timeout 900 python bench.py > gpurun_out/f/bench.json 2> gpurun_out/f/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/f/bench.json'))
print("ms/step",d["ms_per_step"],"value %.3g"%d["value"],"launch us",d["roofline"]["avg_launch_us"])
e=d["extra"]
print("live",d["live_topology"]["ms_per_step"],"single",d["single_mode"]["ms_per_step"],d["single_mode"]["colours"],"world",e["other_configs"]["cfg2_world_step"]["ms_per_step"])
print("cpu",json.dumps(d.get("cpu_baseline"))[:1500])
PY
