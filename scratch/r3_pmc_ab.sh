#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the QP kernel per launch: this build (default layout, AoS results, polish off) and the polish-free build
export TMPDIR=/tmp; ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmcab; rm -rf $OUT; mkdir -p $OUT; cd /tmp
run() { tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $OUT/${tag}_$c -o run -- python $ROOT/bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-batch1 --no-others --no-pmc --no-latency --streams 1 "$@" > $OUT/${tag}_$c.log 2>&1
  done; }
run cur_soa
run cur_aos --output-layout aos
export LMPC_HIP_LIBRARY=$ROOT/racing-lmpc-ros2_amd/lib/liblmpc_hip_nopolish.so
run np_soa
run np_aos --output-layout aos
cd $ROOT
python - <<'PY'
import sqlite3, glob
for d in sorted(glob.glob("gpurun_out/pmcab/*_SIZE")):
    con = sqlite3.connect(glob.glob(d + "/**/run_results.db", recursive=True)[0]); cur = con.cursor()
    for r in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        if "solve" in r[0] or "cleanup" in r[0]: print(d.split("/")[-1], r[0][:48], r[1], "%.1f MB" % (r[2] * 1024 / 1e6), r[3])
PY
rm -rf $OUT/*_SIZE
