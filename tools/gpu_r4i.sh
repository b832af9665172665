#!/bin/bash
# round 4, session i: the half-width island shape (two units per lane) — parity, cfg 2 bench A/B, cfg 4 / cfg 5 kernel times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
export PYTHONUNBUFFERED=1
PHX_EXP_HALF_WIDTH=1 timeout 1500 python -m pytest tests/test_solver_gpu.py tests/test_world_gpu.py -m gpu -x -q -k "not launches_its_own and not config3_size and not 500k and not 200k" 2>&1 | tail -5
for v in base half base half; do
  if [ $v = half ]; then export PHX_EXP_HALF_WIDTH=1; else unset PHX_EXP_HALF_WIDTH; fi
  timeout 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r4i/bench_$v.json 2> gpurun_out/r4i/bench_$v.err
  python - $v <<'PY'
import json,sys
d=json.load(open('gpurun_out/r4i/bench_%s.json'%sys.argv[1]))
print(sys.argv[1],"ms/step",round(d["ms_per_step"],4),"value %.4g"%d["value"],"launch us",round(d["roofline"]["avg_launch_us"],2))
PY
  echo $v cfg5 $(timeout 300 python tools/prof_cfg.py cfg5 2>&1 | tail -1)
  echo $v cfg4 $(timeout 300 python tools/big_world.py 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-300)
done
