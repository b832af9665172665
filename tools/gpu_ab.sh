#!/bin/bash
# A/B of the readback mailbox: world step and bench with and without it, then the GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
export PYTHONUNBUFFERED=1
for i in 1 2; do
  echo "mailbox:";  timeout 300 python tools/world_quick.py 2>&1 | tail -1
  echo "dma:";      PHX_NO_MAILBOX=1 timeout 300 python tools/world_quick.py 2>&1 | tail -1
done
timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/ab/bench.json 2> gpurun_out/ab/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/ab/bench.json'))
print("ms/step",d["ms_per_step"],"value %.3g"%d["value"],"launch us",d["roofline"]["avg_launch_us"])
PY
PHX_NO_MAILBOX=1 timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/ab/bench_dma.json 2> gpurun_out/ab/bench_dma.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/ab/bench_dma.json'))
print("dma: ms/step",d["ms_per_step"],"value %.3g"%d["value"],"launch us",d["roofline"]["avg_launch_us"])
PY
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/ab/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/ab/pytest.log
