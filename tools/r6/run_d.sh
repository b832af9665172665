#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6d}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_world_gpu.py tests/test_c_example.py tests/test_solver_gpu.py -m gpu -x -q -k "reslab or sharded or slab or static_tag or example" > $O/pytest_reslab.log 2>&1; tail -15 $O/pytest_reslab.log
