#!/bin/bash
# HBM traffic of the safe-set kernel (run on the GPU box from the repo root)
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_ss; mkdir -p $OUT; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/$c -o run -- python $ROOT/bench.py --workload lmpc --steps 5 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/$c.log 2>&1
  python3 - <<PY
import sqlite3
con = sqlite3.connect("$OUT/$c/run_results.db")
for r in con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
    if "ss_query" in r[0]: print(r[0][:30], r[1], r[2])
PY
  rm -rf $OUT/$c
done
