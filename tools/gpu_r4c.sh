#!/bin/bash
# round 4, session c: the wave-per-group island kernel (PHX_WAVE_ISLANDS=rows) — parity + A/B on the cfg-2 bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
export PYTHONUNBUFFERED=1
PHX_WAVE_ISLANDS=6 timeout 900 python -m pytest tests/test_solver_gpu.py -m gpu -x -q 2>&1 | tail -5
for v in base wave6 base wave6; do
  if [ $v = wave6 ]; then export PHX_WAVE_ISLANDS=6; else unset PHX_WAVE_ISLANDS; fi
  timeout 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r4c/bench_$v.json 2> gpurun_out/r4c/bench_$v.err
  python - $v <<'PY'
import json,sys
d=json.load(open('gpurun_out/r4c/bench_%s.json'%sys.argv[1]))
print(sys.argv[1],"ms/step",round(d["ms_per_step"],4),"value %.4g"%d["value"],"launch us",round(d["roofline"]["avg_launch_us"],2),"colours",d["config"]["colours"])
PY
done
