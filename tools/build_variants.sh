#!/bin/bash
# builds phyx_amd/lib_v0.so (current flags) and lib_v1.so, lib_v2.so ... = the library with islands.hip recompiled under extra flags
# usage: tools/build_variants.sh "-DPHX_EXP_A" "-DPHX_EXP_B -mllvm ..."
cd "$(dirname "$0")/.."
python -m phyx_amd.build >/dev/null || exit 1
cp phyx_amd/libphyx_amd.so phyx_amd/lib_v0.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -fno-slp-vectorize -Wno-unused-function -mllvm -amdgpu-sched-strategy=iterative-ilp"
i=1
for extra in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $extra -c -o /tmp/islands_v$i.o phyx_amd/csrc/islands.hip || exit 1
  objs=$(ls phyx_amd/build/*.o | grep -v islands.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -ldl -o phyx_amd/lib_v$i.so $objs /tmp/islands_v$i.o || exit 1
  echo "lib_v$i.so: $extra"
  i=$((i+1))
done
