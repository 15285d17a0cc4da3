#!/bin/bash
# round 4, GPU call 37: the fp64 second pass at the top issue priority throughout
python -m pytest tests/test_gpu_mixed_lmpc.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -1
b() { python bench.py "$@" --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value']/1e6,3), round(d['ms_per_step'],4), 'one stream', round(d['ms_per_step_one_stream'],4))"; }
for r in 1 2; do
b --workload iac --horizon 40 --batch 8192 --precision mixed --steps 30
b --workload iac --horizon 40 --batch 8192 --precision f32 --steps 30
done
b --workload lmpc --batch 32768 --precision mixed --regression --steps 10
b --workload lmpc --batch 32768 --precision mixed --steps 10
b --steps 60
