#!/bin/bash
# rocprofv3 kernel-trace stats (CSV) of bench.py; then PMC passes (FETCH_SIZE, WRITE_SIZE) in separate runs.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
timeout 900 python $R/bench.py --steps 20 --warmup 3 > $O/bench_plain.json 2> $O/bench_plain.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o trace -- python $R/bench.py $ARGS > $O/bench_trace.json 2> $O/trace.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o pmc_fetch -- python $R/bench.py $ARGS > $O/bench_pmc_fetch.json 2> $O/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o pmc_write -- python $R/bench.py $ARGS > $O/bench_pmc_write.json 2> $O/pmc_write.err
ls $O
head -12 $O/trace_kernel_stats.csv
cat $O/bench_plain.json
