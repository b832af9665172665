#!/bin/bash
# the flattening pass in two launches: step times (cfg 2, cfg 4, settled), timelines, the GPU suite, fuzz
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6i}; mkdir -p $O
cd $R
timeout 300 python tools/world_quick.py 20 > $O/world_quick.log 2>&1; tail -2 $O/world_quick.log
timeout 300 python tools/r6/settled.py 66 > $O/settled.log 2>&1; tail -1 $O/settled.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('bench', d['ms_per_step'], d['live_topology_ms_per_step'], d['world_step_ms_per_step'])"
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python tools/fuzz.py 660000 1500 > $O/fuzz_small.log 2>&1; tail -1 $O/fuzz_small.log
timeout 900 python tools/fuzz.py 670000 60 --big > $O/fuzz_big.log 2>&1; tail -1 $O/fuzz_big.log
for s in stack merge falling tilted; do echo $s $(python tools/build_twin.py $s 45) $(PHX_NO_PRELABEL=1 python tools/build_twin.py $s 45); done > $O/twins.log 2>&1; cat $O/twins.log
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
