#!/bin/bash
# k_solve_tail: the 200k world's step series (with PHX_NO_TAIL=1 beside), the solver / world GPU tests, big-world fuzz
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6k}; mkdir -p $O
cd $R
timeout 600 python tools/r6/series.py 120 > $O/series.log 2>&1; awk 'NR%5==0' $O/series.log | cut -c1-110 | tail -22
PHX_NO_JP_WALK_ONE=1 timeout 600 python tools/r6/series.py 120 > $O/series_nowalk.log 2>&1; awk "NR%10==0" $O/series_nowalk.log | cut -c1-110 | tail -8
timeout 1800 python -m pytest tests/test_solver_gpu.py tests/test_world_gpu.py -m gpu -x -q > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
timeout 900 python tools/fuzz.py 680000 120 --big > $O/fuzz_big.log 2>&1; tail -1 $O/fuzz_big.log
timeout 900 python tools/fuzz.py 690000 1500 > $O/fuzz_small.log 2>&1; tail -1 $O/fuzz_small.log
timeout 900 python tools/soak.py --long > $O/soak.log 2>&1; tail -2 $O/soak.log
