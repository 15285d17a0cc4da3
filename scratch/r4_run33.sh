#!/bin/bash
# round 4, GPU call 33: k-NN distance pass with four loads in flight per lane and no division
python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "ss_query or safe_set or lmpc or learning" 2>&1 | tail -2
python -m pytest tests/test_gpu_mixed_lmpc.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -1
for a in "--workload lmpc --steps 40" "--workload lmpc --batch 32768 --precision mixed --regression --steps 10"; do
python bench.py $a --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), round(d['ms_per_step'],4), 'one stream', round(d['ms_per_step_one_stream'],4), d['kernels_ms'])"
done
