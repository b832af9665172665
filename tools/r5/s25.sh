#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s; rm -rf $O; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_solver_gpu.py -m gpu -q -x --deselect tests/test_solver_gpu.py::test_island_groups_structure > $O/solver_tests.txt 2>&1; echo "pytest rc $?" >> $O/solver_tests.txt; tail -4 $O/solver_tests.txt
timeout 200 python tools/r5/ab_iters.py 2>&1 | head -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o world -- python $R/tools/world_quick.py 12 > $O/world_quick.txt 2> $O/world.err
cat $O/world_quick.txt
python - <<P
import csv
rows=list(csv.DictReader(open('$O/world_kernel_stats.csv')))
for r in rows[:12]:
    print('%-70s calls %5s avg %8.1f us  pct %s' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
cd $R; timeout 300 python tools/world_quick.py 30
