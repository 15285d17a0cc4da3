#!/bin/bash
# round 5, GPU call 16: the default bench line with its closed-loop leg; kernel trace of the three-launch period
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
( time python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
ROOT=$(pwd)
OUTMD=$ROOT/gpurun_out/r05_closed_loop_fused_rocprof_summary.md
echo "# rocprofv3 --kernel-trace --stats -- python scratch/r5/loop_prof.py <cars> <periods> <cold|warm> [N]  (eager periods; lmpc_loop_advance_batch behind the solve)" > $OUTMD
for cfg in "4096 300 cold" "4096 300 warm" "16384 100 warm"; do
  set -- $cfg
  OUT=/tmp/loopprof2_$1_$3
  rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $ROOT/scratch/r5/loop_prof.py $cfg > $OUT/log.txt 2>&1 )
  echo -e "\n## $cfg\n" >> $OUTMD
  grep "^cars" $OUT/log.txt >> $OUTMD
  python - $OUT >> $OUTMD <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/run_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
print("\n| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
for r in list(cur.execute("select * from top_kernels"))[:6]:
    print("| %s | %d | %.1f | %.3f | %.2f |" % (r[0][:80], r[1], r[2], r[3], r[4]))
PY
done
tail -3 gpurun_out/r05_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_default.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["ms_per_step"], d["ms_per_step_one_stream"], d.get("kernels_ms"), d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"])
print(json.dumps(d.get("closed_loop"), indent=1))
PY
cut -c1-170 $OUTMD
