#!/bin/bash
# fuzz + soak on the restructured rebuild, cfg 4 step time without the profiler, world quick
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6c}; mkdir -p $O
cd $R
timeout 300 python tools/world_quick.py 20 > $O/world_quick.log 2>&1; tail -3 $O/world_quick.log
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
timeout 1200 python tools/fuzz.py 600000 2500 > $O/fuzz_small.log 2>&1; tail -2 $O/fuzz_small.log
timeout 900 python tools/fuzz.py 610000 60 --big > $O/fuzz_big.log 2>&1; tail -2 $O/fuzz_big.log
timeout 900 python tools/soak.py --long > $O/soak.log 2>&1; tail -6 $O/soak.log
for s in stack merge falling tilted; do echo $s $(python tools/build_twin.py $s 45) $(PHX_NO_PRELABEL=1 python tools/build_twin.py $s 45); done > $O/twins.log 2>&1; cat $O/twins.log
