#!/bin/bash
# full bench (secondary measurements included) + world-step timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
export PYTHONUNBUFFERED=1
timeout 900 python bench.py > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3b/bench.json'))
print("ms/step",d["ms_per_step"],"value %.3g"%d["value"],"launch us",d["roofline"]["avg_launch_us"])
e=d["extra"]
print("live",d["live_topology"]["ms_per_step"],"single",d["single_mode"]["ms_per_step"],d["single_mode"]["colours"])
oc=e["other_configs"]
print("world",oc["cfg2_world_step"])
print("cfg4",oc["cfg4_broadphase_1M"])
print("cfg5",oc["cfg5_500k_tall_50it_fp32"])
print("1ofN",e["cfg3_one_rank_of_n"])
print("4x",e["four_times_the_world_one_rank_of_n"])
print("lat",d["roofline"].get("latency_model"))
PY
tail -5 gpurun_out/r3b/bench.err
PHX_WAIT_CLOCK=1 timeout 300 python tools/world_quick.py 2>&1 | tail -5
timeout 300 python tools/big_world.py 2>&1 | tail -15
