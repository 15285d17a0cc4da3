#!/bin/bash
# round 5, GPU call 5: explicit C_A^-1 (MA_MAX = 6) timing + A/B of the fp32 polish's step rule
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
L=$PWD/racing-lmpc-ros2_amd/lib
for v in "" _s4t3e5 _s4t1e5; do
  export LMPC_HIP_LIBRARY=$L/liblmpc_hip$v.so
  echo "=== liblmpc_hip$v.so"
  python tests/dispatch_sweep.py --problems 1024 --families iac > gpurun_out/r5e_sweep_iac$v.txt 2>&1
  python tests/dispatch_sweep.py --problems 1024 --families lrn96,lrn160 --nmax 23 > gpurun_out/r5e_sweep_lrn$v.txt 2>&1
  tail -1 gpurun_out/r5e_sweep_iac$v.txt | cut -c1-900; tail -1 gpurun_out/r5e_sweep_lrn$v.txt | cut -c1-900
  for w in "lmpc_b32768_mixed --workload lmpc --batch 32768 --precision mixed --steps 10" "lmpc_b32768_mixed_regression --workload lmpc --batch 32768 --precision mixed --regression --steps 10" "iac_n40_f32 --workload iac --horizon 40 --batch 8192 --precision f32 --steps 20" "iac_n40_mixed --workload iac --horizon 40 --batch 8192 --precision mixed --steps 20" "lmpc --workload lmpc" "lmpc_b32768 --workload lmpc --batch 32768 --steps 10"; do
    set -- $w; name=$1; shift
    python bench.py "$@" --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'one stream', round(d.get('ms_per_step_one_stream',0),3), d.get('kernels_ms'), 'solved', d['config'].get('solved_fraction'))"
  done
done
unset LMPC_HIP_LIBRARY
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_mixed_lmpc.py tests/test_gpu_dense_fixtures.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -30
