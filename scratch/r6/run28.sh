cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r06_gpu_suite.txt
tail -n 30 gpurun_out/r06_gpu_suite.txt
for a in "" "--horizon 40 --steps 20" "--horizon 60 --steps 10" "--horizon 80 --steps 10" "--workload lmpc"; do python bench.py $a --no-others --no-cpu-baseline --no-pmc --no-batch1 --no-latency 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], '%.4g' % d['value'], 'one stream %.4g' % d['value_one_stream'], d['kernels_ms'], 'iters %.3f' % d['mean_ipm_iters'])"; done
