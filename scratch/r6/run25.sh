# round 6, GPU call 25: default bench line, every bench line, rocprofv3 trace + counter summaries of six configurations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; echo "bench rc $?"
bash scratch/r6/bench_lines.sh > gpurun_out/r06_bench_lines.txt 2>&1
python - <<'PY'
import numpy as np
z = np.load("tests/golden/spec_laps.npz")
np.savez("gpurun_out/spec_laps_only.npz", **{k: z[k] for k in z.files if k.startswith("lap")})
PY
bash scratch/r6/prof.sh tracking > /dev/null 2>&1
bash scratch/r6/prof.sh n60 --horizon 60 --steps 10 > /dev/null 2>&1
bash scratch/r6/prof.sh n80 --horizon 80 --steps 10 > /dev/null 2>&1
bash scratch/r6/prof.sh n40 --horizon 40 --steps 10 > /dev/null 2>&1
bash scratch/r6/prof.sh lmpc --workload lmpc --laps-npz $GRAFT_REPO_ROOT/gpurun_out/spec_laps_only.npz > /dev/null 2>&1
bash scratch/r6/prof.sh iacf32 --workload iac --horizon 40 --batch 8192 --precision f32 > /dev/null 2>&1
cat gpurun_out/r06_bench_lines.txt | tail -n 16
