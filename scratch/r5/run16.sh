#!/bin/bash
# round 5, GPU call 17: the knot-parallel loop tail -- its tests, the loop tests of the suite, the soak, a kernel trace
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/r05_gpu_suite.txt
python scratch/r5/soak_fused.py 2>&1 | grep -v amdgpu > gpurun_out/r05_closed_loop_fused.txt
ROOT=$(pwd)
OUTMD=$ROOT/gpurun_out/r05_closed_loop_fused_rocprof_summary.md
echo "# rocprofv3 --kernel-trace --stats -- python scratch/r5/loop_prof.py <cars> <periods> <cold|warm> [N]  (eager periods; lmpc_loop_advance_batch behind the solve)" > $OUTMD
for cfg in "4096 300 warm" "16384 100 warm" "4096 100 warm 60"; do
  set -- $cfg
  OUT=/tmp/loopprof3_$1_$3_${4:-20}
  rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $ROOT/scratch/r5/loop_prof.py $cfg > $OUT/log.txt 2>&1 )
  echo -e "\n## $cfg\n" >> $OUTMD
  grep "^cars" $OUT/log.txt >> $OUTMD
  python - $OUT >> $OUTMD <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/run_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
print("\n| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
for r in list(cur.execute("select * from top_kernels"))[:4]:
    print("| %s | %d | %.1f | %.3f | %.2f |" % (r[0][:80], r[1], r[2], r[3], r[4]))
PY
done
cat gpurun_out/r05_gpu_suite.txt; cat gpurun_out/r05_closed_loop_fused.txt; cut -c1-170 $OUTMD
