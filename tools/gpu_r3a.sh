#!/bin/bash
# round 3, first check of the resident-layout solver core: parity suite, then bench with and without the fused verification
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3a/pytest.log
tail -25 gpurun_out/r3a/pytest.log
for v in 0 1; do
PHX_NO_FUSED_VERIFY=$v timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r3a/bench_$v.json 2> gpurun_out/r3a/bench_$v.err; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r3a/bench_$v.json'))
    print("no_fused=$v ms/step",d["ms_per_step"],"value %.3g"%d["value"],"launch us",d["roofline"]["avg_launch_us"])
except Exception as e:
    print("bench failed", e); print(open('gpurun_out/r3a/bench_$v.err').read()[-2000:])
PY
done
