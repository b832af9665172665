#!/bin/bash
# bench only (island launch, Single mode, live topology), twice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
export PYTHONUNBUFFERED=1
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/q/bench.json 2> gpurun_out/q/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/q/bench.json'))
e=d["extra"]
print("ms/step",round(d["ms_per_step"],4),"value %.4g"%d["value"],"launch us",round(d["roofline"]["avg_launch_us"],2), "single", round(d["single_mode"]["ms_per_step"],3), "live", round(d["live_topology"]["ms_per_step"],3))
PY
done
