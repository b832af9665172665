#!/bin/bash
# comp-count table with a used list, deferred frontier look: selected tests, cfg 2 / cfg 4 / settled step times, traces of cfg 4 and the settled step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6f}; mkdir -p $O
cd $R
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 python tools/world_quick.py 20 -v > $O/world_quick.log 2>&1; tail -3 $O/world_quick.log
timeout 300 python tools/r6/settled.py 66 > $O/settled.log 2>&1; tail -2 $O/settled.log
PHX_NO_JP_DEFER=1 timeout 300 python tools/r6/settled.py 66 > $O/settled_nodefer.log 2>&1; tail -2 $O/settled_nodefer.log
timeout 300 python - > $O/cfg4_steps.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(10000, 100))
cfg = Configuration(2, 2, 20, 20)
t = []
for step in range(12):
    t0 = time.perf_counter(); w.Update(1/60, cfg); w.sync(); t.append(time.perf_counter() - t0)
print("cfg4 steps ms:", " ".join("%.3f" % (1e3 * x) for x in t), "builds", w.build_counts())
PY
tail -2 $O/cfg4_steps.log
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for c in cfg4; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o ${c}_trace -- python $R/tools/prof_cfg.py $c > $O/${c}_trace.log 2>&1
  python $R/tools/timeline.py $O/${c}_trace_kernel_trace.csv k_keys_buckets -v > $O/${c}_step_timeline.txt 2>&1
done
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o settled_trace -- python $R/tools/r6/settled.py 60 > $O/settled_trace.log 2>&1
python $R/tools/timeline.py $O/settled_trace_kernel_trace.csv k_keys_buckets -v > $O/settled_step_timeline.txt 2>&1
rm -f $O/*_agent_info.csv $O/settled_trace_kernel_trace.csv $O/cfg4_trace_kernel_trace.csv
