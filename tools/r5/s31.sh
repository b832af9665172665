#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5u; rm -rf $O; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
cp $R/phyx_amd/libphyx_amd.so /tmp/keep.so
for V in new v1; do
  if [ $V = v1 ]; then cp $R/_slp/libv1.so $R/phyx_amd/libphyx_amd.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o w_$V -- python $R/tools/world_quick.py 12 > $O/wq_$V.txt 2> $O/w_$V.err
  echo "== $V"; head -1 $O/wq_$V.txt
  python - <<P
import csv
rows=list(csv.DictReader(open('$O/w_${V}_kernel_stats.csv')))
for r in rows:
    if any(k in r['Name'] for k in ('k_build_bin','k_solve_islands')):
        print('%-50s calls %5s avg %8.1f us' % (r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))
P
done
cp /tmp/keep.so $R/phyx_amd/libphyx_amd.so
