#!/bin/bash
# rocprofv3 kernel trace of the eager closed loop: per-kernel durations of a period, cold and warm (GPU box, repo root)
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out
OUTMD=$ROOT/gpurun_out/r05_closed_loop_rocprof_summary.md
echo "# rocprofv3 --kernel-trace --stats -- python scratch/r5/loop_prof.py <cars> <periods> <cold|warm> [N]  (eager periods: one launch per kernel)" > $OUTMD
for cfg in "4096 300 cold" "4096 300 warm" "16384 100 cold" "16384 100 warm" "4096 100 cold 60" "4096 100 warm 60"; do
  set -- $cfg
  OUT=/tmp/loopprof_$1_$3_${4:-20}
  rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $ROOT/scratch/r5/loop_prof.py $cfg > $OUT/log.txt 2>&1 )
  echo -e "\n## $cfg\n" >> $OUTMD
  grep "^cars" $OUT/log.txt >> $OUTMD
  python - $OUT >> $OUTMD <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/run_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
print("\n| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
for r in list(cur.execute("select * from top_kernels"))[:9]:
    print("| %s | %d | %.1f | %.3f | %.2f |" % (r[0][:80], r[1], r[2], r[3], r[4]))
PY
done
cat $OUTMD
