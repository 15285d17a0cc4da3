#!/bin/bash
# round 4, GPU call 25: how many streams the batch pipeline wants (headline, configs[4], N = 60)
mkdir -p gpurun_out
for s in 2 3 4 6; do
  python bench.py --streams $s --steps 40 --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tracking streams $s', round(d['value']/1e6,3), round(d['ms_per_step'],4))"
done
for s in 2 3 4; do
  python bench.py --streams $s --workload lmpc --batch 32768 --precision mixed --regression --steps 10 --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs4 streams $s', round(d['value']/1e6,3), round(d['ms_per_step'],4))"
  python bench.py --streams $s --horizon 60 --steps 10 --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n60 streams $s', round(d['value']/1e6,3), round(d['ms_per_step'],4))"
done
