#!/bin/bash
# round 4: builds (here, no GPU) the library variants of the A/B matrix and of the <double, 7, 0> miscompute bisection
cd "$(dirname "$0")/../racing-lmpc-ros2_amd/csrc" || exit 1
mkdir -p ../lib/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -shared"
b() { name=$1; shift; /opt/rocm/bin/hipcc $F "$@" -o ../lib/ab/liblmpc_$name.so lmpc_lib.hip 2>&1 | grep -E "error" ; }
# polish {inline, call} x FRESH_LANE {off, on}, every instantiation the same way
b inl_nf -DLMPC_POLISH_CALL=0 -DLMPC_FRESH_POLICY=0 &
b call_nf -DLMPC_POLISH_CALL=1 -DLMPC_FRESH_POLICY=0 &
b inl -DLMPC_POLISH_CALL=0 -DLMPC_FRESH_POLICY=1 &
b call -DLMPC_POLISH_CALL=1 -DLMPC_FRESH_POLICY=1 &
wait
# the failing combination (inline polish + FRESH_LANE everywhere: <double, 7, 0> wrong), one sweep function at a time
R="-DLMPC_POLISH_CALL=0 -DLMPC_FRESH_POLICY=1"
b rc_m01 $R -DLMPC_FRESH_MASK=0x01 &
b rc_m08 $R -DLMPC_FRESH_MASK=0x08 &
b rc_m10 $R -DLMPC_FRESH_MASK=0x10 &
b rc_m09 $R -DLMPC_FRESH_MASK=0x09 &
wait
b rc_m18 $R -DLMPC_FRESH_MASK=0x18 &
b rc_m11 $R -DLMPC_FRESH_MASK=0x11 &
b rc_wait0 $R -mllvm -amdgpu-waitcnt-forcezero &
b rc_nosgpr $R -mllvm -amdgpu-spill-sgpr-to-vgpr=false &
wait
b rc_nopost $R -mllvm -enable-post-misched=false &
b rc_O2 $R -O2 &
wait
ls -la ../lib/ab/
