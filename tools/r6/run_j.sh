#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6j}; mkdir -p $O
cd $R
timeout 300 python tools/world_quick.py 20 > $O/world_quick.log 2>&1; tail -2 $O/world_quick.log
timeout 300 python tools/r6/settled.py 66 > $O/settled.log 2>&1; tail -1 $O/settled.log
timeout 900 python -m pytest tests/test_world_gpu.py -m gpu -x -q > $O/pytest_world.log 2>&1; tail -2 $O/pytest_world.log
cd /tmp && export TMPDIR=/tmp
for c in cfg2w cfg4; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o ${c}_trace -- python $R/tools/prof_cfg.py $c > $O/${c}_trace.log 2>&1
  python $R/tools/timeline.py $O/${c}_trace_kernel_trace.csv k_keys_buckets -v > $O/${c}_step_timeline.txt 2>&1
  grep -E "step span|k_cc_compress|k_cc_link" $O/${c}_step_timeline.txt | cut -c1-100
done
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o settled_trace -- python $R/tools/r6/settled.py 60 > $O/settled_trace.log 2>&1
python $R/tools/timeline.py $O/settled_trace_kernel_trace.csv k_keys_buckets -v > $O/settled_step_timeline.txt 2>&1
grep -E "step span|k_cc_compress|k_cc_link" $O/settled_step_timeline.txt | cut -c1-100
rm -f $O/*_agent_info.csv $O/*_kernel_trace.csv
