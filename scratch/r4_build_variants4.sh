#!/bin/bash
# round 4, fourth A/B: the predictor's backward sweep fused into the factorisation (fat layout) against the separate sweep
cd "$(dirname "$0")/../racing-lmpc-ros2_amd/csrc" || exit 1
mkdir -p ../lib/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I."
b() {
  name=$1; shift
  /opt/rocm/bin/hipcc $F "$@" -c -o ../lib/ab/$name.o lmpc_lib.hip 2>&1 | grep -E "error"
  /opt/rocm/bin/hipcc $F "$@" -mllvm -amdgpu-sched-strategy=iterative-minreg -c -o ../lib/ab/${name}_m.o lmpc_lib_minreg.hip 2>&1 | grep -E "error"
  /opt/rocm/bin/hipcc $F -shared -o ../lib/ab/liblmpc_$name.so ../lib/ab/$name.o ../lib/ab/${name}_m.o && rm -f ../lib/ab/$name.o ../lib/ab/${name}_m.o
}
b unfused -DLMPC_FUSE_BWD=0 &
b fused &
wait
ls -la ../lib/ab/
