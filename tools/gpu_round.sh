#!/bin/bash
# One GPU-box session: smoke, bench, dist-path check, rocprofv3 kernel stats. Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O/prof
cd /tmp && export TMPDIR=/tmp
echo "== smoke"; timeout 300 python $R/__graft_entry__.py --smoke 2>&1 | tail -3
echo "== bench N=1"; timeout 900 python $R/bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 3000 $O/bench_n1.json; tail -5 $O/bench_n1.err
echo "== bench via torch.distributed (1 rank, RCCL barrier path)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 1 --steps 5 --warmup 2 --force-dist --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; tail -c 1500 $O/bench_dist1.json; tail -5 $O/bench_dist1.err
echo "== rocprofv3 kernel stats"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o r01 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof/bench_under_rocprof.json 2> $O/prof/rocprof.err
ls -la $O/prof | head; find $O/prof -name "*stats*" | head
