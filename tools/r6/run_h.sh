#!/bin/bash
# cfg 4 step times + timeline (k_bin_components beside the joint match), cfg 2 quick
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6h}; mkdir -p $O
cd $R
timeout 300 python tools/world_quick.py 20 > $O/world_quick.log 2>&1; tail -2 $O/world_quick.log
timeout 300 python - > $O/cfg4_steps.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np
import phyx_amd
from phyx_amd import scenes, Configuration
w = phyx_amd.World(0, gravity=-200.0); w.add_scene(scenes.stack(10000, 100))
cfg = Configuration(2, 2, 20, 20)
t = []
for step in range(24):
    t0 = time.perf_counter(); w.Update(1/60, cfg); w.sync(); t.append(time.perf_counter() - t0)
print("cfg4 steps ms:", " ".join("%.3f" % (1e3 * x) for x in t), "median of the last 16: %.3f" % (1e3 * float(np.median(t[8:]))), "builds", w.build_counts())
PY
tail -2 $O/cfg4_steps.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o cfg4_trace -- python $R/tools/prof_cfg.py cfg4 > $O/cfg4_trace.log 2>&1
python $R/tools/timeline.py $O/cfg4_trace_kernel_trace.csv k_keys_buckets -v > $O/cfg4_step_timeline.txt 2>&1
grep -E "step span|k_bin_components|k_joints_match|k_manifold_slots|k_manifold_components|k_cc_compress" $O/cfg4_step_timeline.txt | cut -c1-110
rm -f $O/*_agent_info.csv $O/cfg4_trace_kernel_trace.csv
