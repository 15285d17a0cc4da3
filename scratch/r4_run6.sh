#!/bin/bash
# round 4, final check as the driver runs it: GPU suite, smoke, bench line
mkdir -p gpurun_out
( time python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r4f_pytest.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4f_smoke.log 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r4f_bench.json 2> gpurun_out/r4f_bench.err
tail -3 gpurun_out/r4f_pytest.log; tail -2 gpurun_out/r4f_smoke.log; tail -4 gpurun_out/r4f_bench.err
