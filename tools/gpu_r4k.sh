#!/bin/bash
# round 4, session k: kernel times of the schedule build in the settled world, the cfg 2 world and the cfg 4 world
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
bash tools/gpu_parts_prof.sh 62 > /dev/null 2>&1
echo "== settled"; grep -E "span|k_cc|k_partner|joint_comp" gpurun_out/parts/last_step.txt
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o world -- python $R/tools/steady.py 12 --no-phase-timing > $O/world_steady.txt 2> $O/world.err
python $R/tools/timeline.py $O/world_kernel_trace.csv k_keys_buckets -v > $O/world_step_timeline.txt 2>&1
echo "== cfg2 world"; grep -E "span|k_cc|k_partner|joint_comp" $O/world_step_timeline.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o cfg4_trace -- python $R/tools/prof_cfg.py cfg4 > $O/cfg4_trace.txt 2> $O/cfg4_trace.err
python $R/tools/timeline.py $O/cfg4_trace_kernel_trace.csv k_keys_buckets -v > $O/cfg4_step_timeline.txt 2>&1
echo "== cfg4 world"; grep -E "span|k_cc|k_partner|joint_comp" $O/cfg4_step_timeline.txt
