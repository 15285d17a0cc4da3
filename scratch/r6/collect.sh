#!/bin/bash
# copy what run25.sh left in gpurun_out/ into profiles/ under the round's names
cd /root/repo
for f in gpurun_out/r06_bench_*.json; do cp $f profiles/$(basename $f); done
cp gpurun_out/r06_bench_lines.txt profiles/r06_bench_lines.txt
for t in tracking n40 n60 n80 lmpc iacf32; do
  [ -f gpurun_out/prof_$t/summary.md ] && cp gpurun_out/prof_$t/summary.md profiles/r06_${t}_rocprof_summary.md
  [ -f gpurun_out/prof_$t/pmc.json ] && cp gpurun_out/prof_$t/pmc.json profiles/r06_pmc_$t.json
done
ls profiles | grep -c r06
