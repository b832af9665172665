#!/bin/bash
# k_solve_parts_ahead A/B in the settled world, partition tests, merged-world lockstep, big fuzz
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6g}; mkdir -p $O
cd $R
timeout 300 python tools/r6/settled.py 66 > $O/settled.log 2>&1; tail -1 $O/settled.log
PHX_PARTS_PLAIN=1 timeout 300 python tools/r6/settled.py 66 > $O/settled_plain.log 2>&1; tail -1 $O/settled_plain.log
timeout 1500 python -m pytest tests/test_solver_gpu.py tests/test_world_gpu.py -m gpu -x -q > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log
timeout 900 python tools/fuzz.py 630000 60 --big > $O/fuzz_big.log 2>&1; tail -2 $O/fuzz_big.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o settled_trace -- python $R/tools/r6/settled.py 60 > $O/settled_trace.log 2>&1
python $R/tools/timeline.py $O/settled_trace_kernel_trace.csv k_keys_buckets -v > $O/settled_step_timeline.txt 2>&1
grep -E "k_solve_parts|k_solve_colour|step span" $O/settled_step_timeline.txt | cut -c1-120
rm -f $O/*_agent_info.csv $O/settled_trace_kernel_trace.csv
