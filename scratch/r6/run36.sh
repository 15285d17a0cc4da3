cd $GRAFT_REPO_ROOT
python tests/dispatch_sweep.py --problems 4096 --seed 8 --families lrn160 --nmin 71 --nmax 71 2>&1 | grep -v amdgpu | tail -2 | cut -c1-200
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_suite.txt 2>&1; echo "rc $?" >> gpurun_out/r06_gpu_suite.txt
tail -n 8 gpurun_out/r06_gpu_suite.txt
for a in "--workload lmpc" "--workload lmpc --batch 32768 --precision mixed --regression --steps 10"; do python bench.py $a --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:50], '%.4g' % d['value'], 'one stream %.4g' % d['value_one_stream'], d['kernels_ms'], 'iters %.3f' % d['mean_ipm_iters'], 'solved', d['solved_fraction'])"; done
for sd in 0 7 8; do timeout 1300 python tests/dispatch_sweep.py --problems 4096 --seed $sd --families lrn96,lrn160 > gpurun_out/r06_sweep_lrn_seed$sd.txt 2>&1; echo "seed $sd rc $?"; grep -E "<--|fault" gpurun_out/r06_sweep_lrn_seed$sd.txt | cut -c1-300 | head -8; done
