#!/bin/bash
# round 4, session h: the sweep as a wave-uniform walk over the candidates (scalar loads) — parity + timelines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests/test_broadphase_gpu.py tests/test_world_gpu.py -m gpu -x -q 2>&1 | tail -6
for i in 1 2; do echo $(timeout 300 python tools/world_quick.py 2>&1 | tail -1); done
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4h
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o world -- python $GRAFT_REPO_ROOT/tools/steady.py 12 --no-phase-timing > $O/world_steady.txt 2> $O/world.err
python $GRAFT_REPO_ROOT/tools/timeline.py $O/world_kernel_trace.csv k_keys_buckets -v > $O/world_step_timeline.txt 2>&1
sed -n '1,2p' $O/world_step_timeline.txt; grep -E "k_sweep|k_emit|k_chunk" $O/world_step_timeline.txt | cut -c1-100
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o cfg4 -- python $GRAFT_REPO_ROOT/tools/prof_cfg.py cfg4 > $O/cfg4.txt 2> $O/cfg4.err
python $GRAFT_REPO_ROOT/tools/timeline.py $O/cfg4_kernel_trace.csv k_keys_buckets -v > $O/cfg4_step_timeline.txt 2>&1
sed -n '1,2p' $O/cfg4_step_timeline.txt; grep -E "k_sweep|k_emit|k_chunk" $O/cfg4_step_timeline.txt | cut -c1-100
