# round 6, GPU call 7: the learning problem's warm start (kernel against twin, closed-loop experiment, facade), the status-parity test with the cap at 60
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_warm.py -x -q -s > gpurun_out/r6_t7a.log 2>&1; echo "rc $?" >> gpurun_out/r6_t7a.log
timeout 1200 python -m pytest tests/test_gpu_facade.py -q -s -k "facade or lmpc" > gpurun_out/r6_t7b.log 2>&1; echo "rc $?" >> gpurun_out/r6_t7b.log
timeout 1200 python -m pytest tests/test_gpu_spec_workload.py tests/test_gpu_loop.py -q -s > gpurun_out/r6_t7c.log 2>&1; echo "rc $?" >> gpurun_out/r6_t7c.log
tail -n 25 gpurun_out/r6_t7a.log; tail -n 8 gpurun_out/r6_t7b.log; tail -n 12 gpurun_out/r6_t7c.log
