#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6e}; mkdir -p $O
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
timeout 1500 python -m pytest tests/test_world_gpu.py tests/test_c_example.py tests/test_solver_gpu.py -m gpu -x -q -k "reslab or sharded or slab or static_tag or example or builder or random" > $O/pytest_sel.log 2>&1; tail -15 $O/pytest_sel.log
timeout 600 python tools/fuzz.py 620000 600 > $O/fuzz_small.log 2>&1; tail -1 $O/fuzz_small.log
