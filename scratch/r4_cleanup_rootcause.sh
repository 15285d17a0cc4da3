#!/bin/bash
# builds (here, no GPU) the variants of the second pass that scratch/r4_cleanup_rootcause.py runs on the GPU box
cd "$(dirname "$0")/../racing-lmpc-ros2_amd/csrc" || exit 1
mkdir -p ../lib/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -shared -DLMPC_DEBUG_HOOKS -DLMPC_MIXED_LONG_LEARNING"
/opt/rocm/bin/hipcc $F -o ../lib/ab/liblmpc_rc_call.so lmpc_lib.hip &                                                     # the product form (a call)
/opt/rocm/bin/hipcc $F -DLMPC_CLEANUP_INLINE -o ../lib/ab/liblmpc_rc_inline.so lmpc_lib.hip &                               # the failing form
/opt/rocm/bin/hipcc $F -DLMPC_CLEANUP_INLINE -DLMPC_CLEANUP_LOOP_WAIT -o ../lib/ab/liblmpc_rc_inline_wait.so lmpc_lib.hip & # A: drain everything at the loop top
/opt/rocm/bin/hipcc $F -DLMPC_CLEANUP_INLINE -DLMPC_CLEANUP_LDS_CLEAR -o ../lib/ab/liblmpc_rc_inline_clear.so lmpc_lib.hip &# B: no stale LDS
wait
/opt/rocm/bin/hipcc $F -DLMPC_CLEANUP_INLINE -mllvm -amdgpu-spill-sgpr-to-vgpr=false -o ../lib/ab/liblmpc_rc_inline_nosgprlane.so lmpc_lib.hip &  # C: SGPR spills not in VGPR lanes
/opt/rocm/bin/hipcc $F -DLMPC_CLEANUP_INLINE -DLMPC_NO_FRESH_LANE -DLMPC_POLISH_CALL=0 -o ../lib/ab/liblmpc_rc_inline_r3like.so lmpc_lib.hip &  # the round-3 code shape
wait
ls -la ../lib/ab/liblmpc_rc_*
