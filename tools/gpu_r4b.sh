#!/bin/bash
# round 4, session b: A/B of the windowed colouring experiment (PHX_EXP_WINDOW) on the cfg-2 bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
export PYTHONUNBUFFERED=1
for v in base exp base exp; do
  if [ $v = exp ]; then export PHX_EXP_WINDOW=1; else unset PHX_EXP_WINDOW; fi
  timeout 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r4b/bench_$v.json 2> gpurun_out/r4b/bench_$v.err
  python - $v <<'PY'
import json,sys
d=json.load(open('gpurun_out/r4b/bench_%s.json'%sys.argv[1]))
print(sys.argv[1],"ms/step",round(d["ms_per_step"],4),"value %.4g"%d["value"],"launch us",round(d["roofline"]["avg_launch_us"],2),"colours",d["config"]["colours"],d["roofline"].get("latency_model"))
PY
done
