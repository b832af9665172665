#!/bin/bash
# the last session: the GPU suite and both bench lines on the final code with the final profiles on record
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6v2}; mkdir -p $O
cd $R
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 300 $O/bench_default.json; echo
timeout 900 python bench.py --secondary > $O/bench_plain.json 2> $O/bench_plain.err; tail -2 $O/bench_plain.err
timeout 300 python tools/world_quick.py 20 > $O/world_quick.log 2>&1; tail -2 $O/world_quick.log
