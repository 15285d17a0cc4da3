#!/bin/bash
# round 5, GPU call 7: safe set by reference + bench hygiene: suite, default bench, learning bench lines (spec / near x idx / arrays)
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
( time python -m pytest tests -q -m gpu -x --durations=5 2>&1 | tail -25 ) > gpurun_out/r5g_pytest.log 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5g_bench.json 2> gpurun_out/r5g_bench.err
line() { name=$1; shift; python bench.py "$@" --no-others --no-cpu-baseline --no-pmc 2>gpurun_out/r5g_$name.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'one stream', round(d.get('ms_per_step_one_stream',0),3), d.get('kernels_ms'), 'solved', d.get('solved_fraction'), 'iters', round(d.get('mean_ipm_iters',0),2), 'window', round(d.get('timed_window_s',0),2), (d.get('ss_query_kernel') or {}).get('ms'))"; }
line lmpc_spec_idx --workload lmpc
line lmpc_spec_arrays --workload lmpc --ss-mode arrays
line lmpc_near_idx --workload lmpc --lmpc-data near
line lmpc_near_arrays --workload lmpc --lmpc-data near --ss-mode arrays
line lmpc32k_near_idx_mixed_reg --workload lmpc --batch 32768 --precision mixed --regression --lmpc-data near --steps 10
line lmpc32k_near_arrays_mixed_reg --workload lmpc --batch 32768 --precision mixed --regression --lmpc-data near --ss-mode arrays --steps 10
line lmpc32k_spec_idx_mixed_reg --workload lmpc --batch 32768 --precision mixed --regression --steps 10
line lmpc32k_spec_idx_f64 --workload lmpc --batch 32768 --steps 10
cat gpurun_out/r5g_pytest.log; tail -1 gpurun_out/r5g_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','ms_per_step','ms_per_step_one_stream','timed_steps','timed_window_s','p99_solve_ms','solved_fraction','mean_ipm_iters','kernels_ms')})
print(d['roofline'])
print(d.get('cpu_baseline',{}).get('value'))
for o in d.get('others',[]): print(o.get('baseline_config'), o.get('value'), o.get('ms_per_step'), o.get('ms_per_step_one_stream'), o.get('solved_fraction'), o.get('error'))
"; tail -3 gpurun_out/r5g_bench.json | head -c 300
