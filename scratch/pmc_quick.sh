export TMPDIR=/tmp; ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/p3 -o run -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batch1 --no-others --no-pmc --streams 1 > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/p4 -o run -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batch1 --no-others --no-pmc --streams 1 > $OUT/p4.log 2>&1
cd $ROOT
python - <<'PY'
import sqlite3
for n in ("p3","p4"):
    con=sqlite3.connect(f"gpurun_out/pmcq/{n}/run_results.db"); cur=con.cursor()
    for r in cur.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        if "solve" in r[0]: print(r[0][:40], r[1], r[2])
PY
rm -rf $OUT/p3 $OUT/p4
