#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5k/bench_default.json'))
print({k:d.get(k) for k in ("value","ms_per_step","live_topology_ms_per_step","world_step_ms_per_step","contacts_resolved_per_sec")})
print(d["extra"].get("result_check") or d.get("result_check"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
