#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6b}; mkdir -p $O
cd $R
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
PHX_TRACE_SPEC=1 timeout 300 python tools/world_quick.py 20 -v > $O/world_quick.log 2>&1; tail -5 $O/world_quick.log
PHX_SWEEP_ROWS=1 timeout 300 python tools/world_quick.py 20 > $O/world_quick_sweeprows.log 2>&1; tail -3 $O/world_quick_sweeprows.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 600 $O/bench_default.json; tail -3 $O/bench_default.err
timeout 300 python tools/r6/settled.py 62 > $O/settled.log 2>&1; tail -2 $O/settled.log
cd /tmp && export TMPDIR=/tmp
for c in cfg2w cfg4; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o ${c}_trace -- python $R/tools/prof_cfg.py $c > $O/${c}_trace.log 2>&1
  python $R/tools/timeline.py $O/${c}_trace_kernel_trace.csv k_keys_buckets -v > $O/${c}_step_timeline.txt 2>&1
done
head -45 $O/cfg2w_step_timeline.txt; head -60 $O/cfg4_step_timeline.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o settled_trace -- python $R/tools/r6/settled.py 60 > $O/settled_trace.log 2>&1
python $R/tools/timeline.py $O/settled_trace_kernel_trace.csv k_keys_buckets -v > $O/settled_step_timeline.txt 2>&1
rm -f $O/*_agent_info.csv $O/settled_trace_kernel_trace.csv $O/cfg4_trace_kernel_trace.csv
