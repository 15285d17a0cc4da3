cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r06_gpu_suite.txt
tail -n 12 gpurun_out/r06_gpu_suite.txt
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g' % d['value'], 'one stream %.4g' % d['value_one_stream'], d['kernels_ms'], 'iters %.3f' % d['mean_ipm_iters'], 'solved', d['solved_fraction'], d['roofline'].get('traffic_over_algorithmic'), d['roofline'].get('issue'), d['closed_loop']['cold']['value'], d['closed_loop']['warm']['value'], d['batch1_solve_ms'])"
python scratch/r5/soak_fused.py 2>&1 | grep -v amdgpu | head -4
