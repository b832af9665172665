#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o world -- python $R/tools/world_quick.py 12 > $O/world_quick.txt 2> $O/world.err
cat $O/world_quick.txt
python - <<P
import csv
rows=list(csv.DictReader(open('$O/world_kernel_stats.csv')))
for r in rows[:24]:
    print('%-70s calls %5s avg %8.1f us  pct %s' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
