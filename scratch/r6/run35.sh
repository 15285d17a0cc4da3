cd $GRAFT_REPO_ROOT
python tests/dispatch_sweep.py --problems 4096 --families lrn160 --nmin 57 --nmax 57 2>&1 | grep -v amdgpu | tail -2 | cut -c1-260
python tests/dispatch_sweep.py --problems 4096 --families trk --nmin 14 --nmax 14 2>&1 | grep -v amdgpu | tail -2 | cut -c1-200
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r06_gpu_suite.txt
tail -n 8 gpurun_out/r06_gpu_suite.txt
python bench.py --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g' % d['value'], 'one stream %.4g' % d['value_one_stream'], d['kernels_ms'], 'iters %.3f' % d['mean_ipm_iters'], 'solved', d['solved_fraction'])"
