#!/bin/bash
# round 5, GPU call 11: the round's record -- bench lines, rocprofv3 trace + PMC of five workloads, full-size parity, sweep, soak, mixed tail
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
python scratch/r5/soak_warm.py 2>&1 | grep -v amdgpu > gpurun_out/r05_closed_loop_soak.txt
python scratch/r5/fullsize_parity.py 2>&1 | grep -v amdgpu > gpurun_out/r05_fullsize_parity.txt
python tests/dispatch_sweep.py --problems 1024 2>&1 | grep -v amdgpu > gpurun_out/r05_dispatch_sweep.txt
python scratch/r5/mixed_tail.py 2>&1 | grep -v amdgpu > gpurun_out/r05_mixed_tail.txt
bash scratch/r5/bench_lines.sh > gpurun_out/r05_bench_lines.txt 2>&1
python bench.py --workload lmpc --lmpc-data near --no-others --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_lmpc_near.json
python bench.py --workload lmpc --lmpc-data near --batch 32768 --precision mixed --regression --steps 10 --no-others --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_bench_lmpc_b32768_mixed_regression_near.json
for w in "tracking" "lmpc --workload lmpc" "iacf32 --workload iac --horizon 40 --batch 8192 --precision f32" "n60 --horizon 60" "lmpcmix --workload lmpc --batch 32768 --precision mixed --regression"; do
  set -- $w; tag=$1; shift
  bash scratch/r5/prof.sh $tag "$@" > gpurun_out/prof_$tag.log 2>&1
  cp gpurun_out/prof_$tag/summary.md gpurun_out/r05_${tag}_rocprof_summary.md 2>/dev/null
  cp gpurun_out/prof_$tag/pmc.json gpurun_out/r05_pmc_${tag}.json 2>/dev/null
done
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
cat gpurun_out/r05_closed_loop_soak.txt; cat gpurun_out/r05_fullsize_parity.txt; tail -1 gpurun_out/r05_dispatch_sweep.txt | cut -c1-700; cut -c1-300 gpurun_out/r05_bench_lines.txt; head -12 gpurun_out/r05_tracking_rocprof_summary.md; tail -3 gpurun_out/r05_bench_default.err
