#!/bin/bash
# round 4, session a: what bounds a class step (tools/probe/unit_issue) + SQ counters of the island kernel at cfg 2
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4a
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 $R/tools/probe/unit_issue > $O/unit_issue.txt 2>&1
rocprofv3 -L > $O/counters.txt 2>&1
ARGS="--steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d $O -o sq1 -- python $R/bench.py $ARGS > $O/bench_sq1.json 2> $O/sq1.err
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $O -o sq2 -- python $R/bench.py $ARGS > $O/bench_sq2.json 2> $O/sq2.err
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O -o sq3 -- python $R/bench.py $ARGS > $O/bench_sq3.json 2> $O/sq3.err
ls -la $O
cat $O/unit_issue.txt
tail -3 $O/sq1.err
