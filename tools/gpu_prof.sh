#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of bench.py (secondary measurements included, so the HBM colour path
# is in it), then PMC passes (FETCH_SIZE, WRITE_SIZE) in separate runs, then a kernel trace of a running world (timeline).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --secondary"
timeout 900 python $R/bench.py --secondary > $O/bench_plain.json 2> $O/bench_plain.err
timeout 600 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o trace -- python $R/bench.py $ARGS > $O/bench_trace.json 2> $O/trace.err
# the main measurement alone (no secondary runs): every launch of the island kernel in this trace is a full-grid, 20-sweep launch,
# so its average is the number bench.py's HIP events must agree with
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o main -- python $R/bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_main.json 2> $O/main.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o pmc_fetch -- python $R/bench.py $ARGS > $O/bench_pmc_fetch.json 2> $O/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o pmc_write -- python $R/bench.py $ARGS > $O/bench_pmc_write.json 2> $O/pmc_write.err
timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O -o world -- python $R/tools/steady.py 12 --no-phase-timing > $O/world_steady.txt 2> $O/world.err
python $R/tools/timeline.py $O/world_kernel_trace.csv k_keys_buckets -v > $O/world_step_timeline.txt 2>&1
ls $O
head -14 $O/trace_kernel_stats.csv
head -3 $O/world_step_timeline.txt
cat $O/bench_plain.json | head -c 600
