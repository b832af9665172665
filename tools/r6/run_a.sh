#!/bin/bash
# round 6, first GPU session: smoke, the cfg 2 world step (spoils traced), the GPU suite, the default bench line, the static-tag deviation,
# and a kernel trace of the running cfg 2 world (timeline of one steady step)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6a}; mkdir -p $O
cd $R
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
PHX_TRACE_SPEC=1 timeout 300 python tools/world_quick.py 20 -v > $O/world_quick.log 2>&1; tail -5 $O/world_quick.log
PHX_TRACE_SPEC=1 PHX_NO_PRELABEL=1 timeout 300 python tools/world_quick.py 20 > $O/world_quick_noprelabel.log 2>&1; tail -3 $O/world_quick_noprelabel.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
timeout 600 python tools/static_tag_deviation.py cfg2 cfg5 --out $O/static_tag_deviation.json > $O/static_tag.log 2>&1; tail -3 $O/static_tag.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o cfg2w_trace -- python $R/tools/prof_cfg.py cfg2w > $O/cfg2w_trace.log 2>&1
python $R/tools/timeline.py $O/cfg2w_trace_kernel_trace.csv k_keys_buckets -v > $O/cfg2w_step_timeline.txt 2>&1; head -60 $O/cfg2w_step_timeline.txt
rm -f $O/*_agent_info.csv
