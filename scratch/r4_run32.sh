#!/bin/bash
# round 4, GPU call 32: stream counts with the co-resident linearisation (headline, IAC fp32 / mixed)
for s in 2 3 4 6 8; do
  python bench.py --streams $s --steps 60 --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tracking streams $s', round(d['value']/1e6,3), round(d['ms_per_step'],4))"
  python bench.py --streams $s --workload iac --horizon 40 --batch 8192 --precision f32 --steps 30 --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('iac f32 streams $s', round(d['value']/1e6,3), round(d['ms_per_step'],4))"
done
for s in 3 4 6; do
  python bench.py --streams $s --workload iac --horizon 40 --batch 8192 --precision mixed --steps 30 --no-others --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('iac mixed streams $s', round(d['value']/1e6,3), round(d['ms_per_step'],4))"
done
