# round 6, GPU call 4: the new parity tests, the fallback, the sharded solver, the preflights, smoke
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_spec_workload.py tests/test_gpu_dense_fixtures.py tests/test_gpu_ss_idx.py -x -q -s > gpurun_out/r6_t4a.log 2>&1; echo "rc $?" >> gpurun_out/r6_t4a.log
timeout 1500 python -m pytest tests/test_gpu_facade.py -k "sharded" -x -q -s > gpurun_out/r6_t4b.log 2>&1; echo "rc $?" >> gpurun_out/r6_t4b.log
timeout 1500 python -m pytest tests/test_gpu_path.py -k "preflight or mixed_precision or hard_convex" -x -q -s > gpurun_out/r6_t4c.log 2>&1; echo "rc $?" >> gpurun_out/r6_t4c.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "rc $?" >> gpurun_out/r6_smoke.log
tail -5 gpurun_out/r6_t4a.log gpurun_out/r6_t4b.log gpurun_out/r6_t4c.log gpurun_out/r6_smoke.log
