#!/bin/bash
# round 4, GPU call 30: pipelined rates with the linearisation kernel at one / two waves per SIMD
mkdir -p gpurun_out
AB=racing-lmpc-ros2_amd/lib/ab
b() { v=$1; shift; LMPC_HIP_LIBRARY=$PWD/$AB/liblmpc_$v.so python bench.py "$@" --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$*', round(d['value']/1e6,3), round(d['ms_per_step'],4), 'one stream', round(d['ms_per_step_one_stream'],4), {k: round(x,4) for k,x in d['kernels_ms'].items()})"; }
for v in lw1 lw2 lw1 lw2; do b $v --steps 40; done
for v in lw1 lw2; do
  b $v --steps 20 --batch 65536
  b $v --workload lmpc --steps 40
  b $v --workload lmpc --batch 32768 --precision mixed --regression --steps 10
  b $v --workload iac --horizon 40 --batch 8192 --steps 20
  b $v --workload iac --horizon 40 --batch 8192 --precision f32 --steps 20
  b $v --horizon 60 --steps 10
  b $v --horizon 40 --steps 20
done
