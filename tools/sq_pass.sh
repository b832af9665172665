#!/bin/bash
# SQ counters of the island kernel (the dominant kernel of cfg 2): three rocprofv3 --pmc passes of the main measurement alone
# (every k_solve_islands launch in it is a full-grid, 20-sweep launch), summarised by tools/sq_summary.py into profiles/<tag>_sq_islands.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sq
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O -o sq1 -- $CMD > $O/sq1.json 2> $O/sq1.err
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $O -o sq2 -- $CMD > $O/sq2.json 2> $O/sq2.err
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O -o sq3 -- $CMD > $O/sq3.json 2> $O/sq3.err
ls $O | head -20
python $R/tools/sq_summary.py $O $O/sq_islands.json
