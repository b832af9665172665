#!/bin/bash
# functional check of bench.py's N=2 path on a 1-GPU box: two processes, both on GPU 0, gloo for the collectives
cd $GRAFT_REPO_ROOT
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 timeout 300 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 3 --columns 400 > gpurun_out/rank$r.log 2>&1 &
done
wait
tail -c 1500 gpurun_out/rank0.log; echo; tail -3 gpurun_out/rank1.log
