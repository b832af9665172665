#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5r; mkdir -p $O
export PYTHONUNBUFFERED=1
PHX_NO_TWINS=1 timeout 900 python bench.py --no-cpu-baseline --secondary > $O/bench_sec.json 2> $O/bench_sec.err; tail -3 $O/bench_sec.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r5r/bench_sec.json'))
print('ms/step', d['ms_per_step'], 'live', d['live_topology_ms_per_step'], 'world', d['world_step_ms_per_step'], 'launch', d['roofline']['avg_launch_us'])
e=d['extra']
for k in ('cfg3_one_rank_of_n','cfg3_slab_one_rank_of_n','four_times_the_world_one_rank_of_n'):
    print(k, {n:(round(v['ms_per_step'],4), round(v['island_launch_us'],1)) for n,v in e[k].items() if isinstance(v,dict)})
print('single', d['single_mode']['ms_per_step'])
oc=e['other_configs']
for k,v in oc.items():
    print(k, {kk:vv for kk,vv in v.items() if isinstance(vv,(int,float))})
P
