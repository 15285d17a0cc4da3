cd $GRAFT_REPO_ROOT
T=tests/golden/barc_track/15_barc_optm.txt
L=racing-lmpc-ros2_amd/lib
for n in 20 60; do $L/test_node_core $T $n 2.1 continuous | tail -2; done
$L/bench_cabi $T 4096 20 | tail -1
$L/bench_cabi $T 4096 20 --gpus 2 --same-device --gather copy | tail -1
$L/bench_cabi $T 4096 10 --horizon 60 2>/dev/null | tail -1
python scratch/r5/host_latency.py 2>&1 | grep -v amdgpu | tail -6
