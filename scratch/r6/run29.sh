cd $GRAFT_REPO_ROOT
for lib in liblmpc_hip_f57.so liblmpc_hip_f57ns.so; do echo "== $lib"; LMPC_HIP_LIBRARY=racing-lmpc-ros2_amd/lib/$lib python scratch/r6/n20_probe.py 2>&1 | grep -v amdgpu | head -3;
LMPC_HIP_LIBRARY=racing-lmpc-ros2_amd/lib/$lib python bench.py --no-others --no-cpu-baseline --no-pmc --no-batch1 --no-latency 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g' % d['value'], 'one stream %.4g' % d['value_one_stream'], d['kernels_ms'], 'iters %.3f' % d['mean_ipm_iters'], 'solved', d['solved_fraction'])"; done
