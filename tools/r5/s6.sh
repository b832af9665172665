#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5f/bench_default.json'))
print({k:d.get(k) for k in ("value","ms_per_step","live_topology_ms_per_step","world_step_ms_per_step","contacts_resolved_per_sec")})
print(d["roofline"]["traffic_source"]); print(d.get("cpu_baseline",{}).get("value"))
PY
bash tools/r5/steady_prof.sh r5f > $O/steady_prof.log 2>&1; tail -80 $O/steady_prof.log
