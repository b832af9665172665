#!/bin/bash
# round 6 profile session 1: the bench's kernel stats + PMC passes (tools/gpu_prof.sh), cfg 5 (side_prof.sh), one steady World::Update of
# cfg 4 and cfg 2 kernel by kernel (r5/steady_prof.sh), the settled world's last step, the island kernel's SQ counters (sq_pass.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/gpu_prof.sh > $R/gpurun_out/gpu_prof.log 2>&1; tail -5 $R/gpurun_out/gpu_prof.log
bash tools/side_prof.sh > $R/gpurun_out/side_prof.log 2>&1
bash tools/r5/steady_prof.sh steady > $R/gpurun_out/steady_prof.log 2>&1; tail -3 $R/gpurun_out/steady_prof.log
O=$R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o settled -- python $R/tools/r6/settled.py 60 > $O/settled.log 2>&1
python $R/tools/timeline.py $O/settled_kernel_trace.csv k_keys_buckets -v > $O/settled_last_step.txt 2>&1; head -3 $O/settled_last_step.txt
cd $R
bash tools/sq_pass.sh > $R/gpurun_out/sq_pass.log 2>&1; tail -3 $R/gpurun_out/sq_pass.log
# keep what travels back small: the raw traces of the long runs are summarised above
rm -f $O/*_agent_info.csv $O/settled_kernel_trace.csv $O/world_kernel_trace.csv $R/gpurun_out/steady/*_kernel_trace.csv $R/gpurun_out/steady/*_counter_collection.csv $R/gpurun_out/sq/*_kernel_trace.csv
du -sh $R/gpurun_out
