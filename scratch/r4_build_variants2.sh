#!/bin/bash
# round 4, second A/B matrix (built here, no GPU): the row phases of the iteration -- slots per chunk, slot tables behind an
# empty asm (no hoisted LDS addresses), fresh lane in the model stream's fetch.  Both translation units per variant.
cd "$(dirname "$0")/../racing-lmpc-ros2_amd/csrc" || exit 1
mkdir -p ../lib/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I."
b() {
  name=$1; shift
  /opt/rocm/bin/hipcc $F "$@" -c -o ../lib/ab/$name.o lmpc_lib.hip 2>&1 | grep -E "error"
  /opt/rocm/bin/hipcc $F "$@" -mllvm -amdgpu-sched-strategy=iterative-minreg -c -o ../lib/ab/${name}_m.o lmpc_lib_minreg.hip 2>&1 | grep -E "error"
  /opt/rocm/bin/hipcc $F -shared -o ../lib/ab/liblmpc_$name.so ../lib/ab/$name.o ../lib/ab/${name}_m.o && rm -f ../lib/ab/$name.o ../lib/ab/${name}_m.o
}
b old -DLMPC_ROW_CHUNK=0 -DLMPC_OPAQUE_MIN_KQ=99 -DLMPC_FETCH_FRESH=0 &
b new &
b c3 -DLMPC_ROW_CHUNK=3 &
wait
b c6 -DLMPC_ROW_CHUNK=6 &
b nofetch -DLMPC_FETCH_FRESH=0 &
b newnc -DLMPC_ROW_CHUNK=0 &
wait
b op7 -DLMPC_OPAQUE_MIN_KQ=7 &
b op7c -DLMPC_OPAQUE_MIN_KQ=7 -DLMPC_ROW_CHUNK_MIN_KQ=7 &
b op4 -DLMPC_OPAQUE_MIN_KQ=2 &
wait
ls -la ../lib/ab/
