#!/bin/bash
# the last validation session of round 6: smoke, the GPU suite, both bench lines, fuzz (small + big), the long soaks, the rebuild twins,
# and the settled world's last step (kernel stats + timeline) for profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6v5}; mkdir -p $O
cd $R
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 250 $O/bench_default.json; echo
timeout 900 python bench.py --secondary > $O/bench_plain.json 2> $O/bench_plain.err; tail -1 $O/bench_plain.err
timeout 1500 python tools/fuzz.py 700000 3000 > $O/fuzz_small.log 2>&1; tail -1 $O/fuzz_small.log
timeout 1200 python tools/fuzz.py 710000 120 --big > $O/fuzz_big.log 2>&1; tail -1 $O/fuzz_big.log
timeout 1200 python tools/soak.py --long > $O/soak.log 2>&1; tail -1 $O/soak.log
for s in stack merge falling tilted; do echo $s $(python tools/build_twin.py $s 45) $(PHX_NO_PRELABEL=1 python tools/build_twin.py $s 45); done > $O/twins.log 2>&1; cat $O/twins.log | cut -c1-100
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o settled -- python $R/tools/r6/settled.py 60 > $O/settled.log 2>&1
python $R/tools/timeline.py $O/settled_kernel_trace.csv k_keys_buckets -v > $O/settled_last_step.txt 2>&1; head -1 $O/settled_last_step.txt
rm -f $O/*_agent_info.csv $O/settled_kernel_trace.csv
