#!/bin/bash
# round 6 validation session on the final code: the GPU suite, differential fuzz (small + big worlds), the long soaks, the rebuild twins,
# and the bench lines (default = what the driver runs; --secondary = the builder's full line) with the r06 PMC traffic on record
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r6v}; mkdir -p $O
cd $R
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 400 $O/bench_default.json; echo
timeout 900 python bench.py --secondary > $O/bench_plain.json 2> $O/bench_plain.err; tail -2 $O/bench_plain.err
timeout 1500 python tools/fuzz.py 640000 4000 > $O/fuzz_small.log 2>&1; tail -1 $O/fuzz_small.log
timeout 1200 python tools/fuzz.py 650000 120 --big > $O/fuzz_big.log 2>&1; tail -1 $O/fuzz_big.log
timeout 1200 python tools/soak.py --long > $O/soak.log 2>&1; tail -6 $O/soak.log
for s in stack merge falling tilted; do echo $s $(python tools/build_twin.py $s 45) $(PHX_NO_PRELABEL=1 python tools/build_twin.py $s 45); done > $O/twins.log 2>&1; cat $O/twins.log
