#!/bin/bash
# kernel-time breakdown of the closed-loop harness (run on the GPU box from the repo root)
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_soak; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- python $ROOT/scratch/soak.py > $OUT/trace.log 2>&1
python3 - <<PY
import sqlite3
con = sqlite3.connect("$OUT/trace/run_results.db")
print("| kernel | calls | total us | avg us | % |")
for r in con.execute("select * from top_kernels"):
    print("| %s | %d | %.1f | %.3f | %.2f |" % (r[0][:60], r[1], r[2], r[3], r[4]))
PY
tail -1 $OUT/trace.log
rm -rf $OUT/trace
