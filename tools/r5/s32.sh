#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r5u; rm -rf $O; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_solver_gpu.py -m gpu -q -x > $O/solver_tests.txt 2>&1; echo "pytest rc $?" >> $O/solver_tests.txt; tail -4 $O/solver_tests.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o w -- python $R/tools/world_quick.py 12 > $O/wq.txt 2> $O/w.err
head -1 $O/wq.txt
python - <<P
import csv
rows=list(csv.DictReader(open('$O/w_kernel_stats.csv')))
for r in rows:
    if any(k in r['Name'] for k in ('k_build_bin','k_solve_islands')):
        print('%-50s calls %5s avg %8.1f us' % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))
P
cd $R; timeout 300 python tools/world_quick.py 30 | head -1; timeout 300 python tools/world_quick.py 30 | head -1
timeout 900 python -m pytest tests/test_world_gpu.py -m gpu -q -x 2>&1 | tail -3
