#!/bin/bash
# round 4, fifth A/B: flag-select addresses recomputed at four sites of the short-horizon tracking kernels (LMPC_OPAQUE_SITES bit mask)
cd "$(dirname "$0")/../racing-lmpc-ros2_amd/csrc" || exit 1
mkdir -p ../lib/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I."
b() {
  name=$1; shift
  /opt/rocm/bin/hipcc $F "$@" -c -o ../lib/ab/$name.o lmpc_lib.hip 2>&1 | grep -E "error"
  /opt/rocm/bin/hipcc $F "$@" -mllvm -amdgpu-sched-strategy=iterative-minreg -c -o ../lib/ab/${name}_m.o lmpc_lib_minreg.hip 2>&1 | grep -E "error"
  /opt/rocm/bin/hipcc $F -shared -o ../lib/ab/liblmpc_$name.so ../lib/ab/$name.o ../lib/ab/${name}_m.o && rm -f ../lib/ab/$name.o ../lib/ab/${name}_m.o
}
b s0 -DLMPC_OPAQUE_SITES=0 &
b s15 -DLMPC_OPAQUE_SITES=15 &
b s13 -DLMPC_OPAQUE_SITES=13 &
wait
b s5 -DLMPC_OPAQUE_SITES=5 &
b s8 -DLMPC_OPAQUE_SITES=8 &
b s7 -DLMPC_OPAQUE_SITES=7 &
wait
ls ../lib/ab/
