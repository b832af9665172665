#!/bin/bash
# round-2 GPU session 1: parity suite, default bench, dist paths on one GPU, island kernel phase trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s1/pytest.log
tail -15 gpurun_out/s1/pytest.log
timeout 600 python bench.py > gpurun_out/s1/bench_n1.json 2> gpurun_out/s1/bench_n1.err; echo "bench rc=$?"
tail -c 600 gpurun_out/s1/bench_n1.err
timeout 300 python bench.py --force-dist --no-secondary --no-cpu-baseline > gpurun_out/s1/bench_dist1.json 2> gpurun_out/s1/bench_dist1.err; echo "bench dist1 rc=$?"
tail -c 600 gpurun_out/s1/bench_dist1.err
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 timeout 300 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 2 --repeats 3 --no-secondary --no-cpu-baseline > gpurun_out/s1/rank$r.log 2>&1 &
done
wait
tail -c 1500 gpurun_out/s1/rank0.log; echo; tail -c 300 gpurun_out/s1/rank1.log
timeout 300 python tools/island_trace.py > gpurun_out/s1/island_trace.txt 2>&1; cat gpurun_out/s1/island_trace.txt
timeout 300 python tools/island_trace.py 1000 200 20 8 > gpurun_out/s1/island_trace_shard8.txt 2>&1; cat gpurun_out/s1/island_trace_shard8.txt
