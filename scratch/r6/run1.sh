# round 6, GPU call 1: the spec laps as data, what the spec LMPC workload looks like per problem, phase split of the kernels at the shipped horizons
mkdir -p gpurun_out
python tests/golden/make_spec_laps.py gpurun_out/spec_laps.npz > gpurun_out/r6_spec_laps.log 2>&1
timeout 900 python scratch/r6/spec_probe.py > gpurun_out/r6_spec_probe.log 2>&1
for n in 20 40 60 80; do timeout 300 python scratch/phase_timing.py $n 4096 > gpurun_out/r6_phase_n$n.log 2>&1; done
timeout 300 python scratch/phase_timing.py 20 4096 lmpc > gpurun_out/r6_phase_lmpc.log 2>&1
tail -3 gpurun_out/r6_spec_laps.log; cat gpurun_out/r6_spec_probe.log | tail -20
