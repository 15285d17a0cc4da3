#!/bin/bash
mkdir -p gpurun_out
AB=$PWD/racing-lmpc-ros2_amd/lib/ab
for rep in 1 2; do
timeout 400 python scratch/r4_ab.py trk40 iac lmpc lmpc96 lmpc40 > gpurun_out/r4j_ab_main_$rep.jsonl 2> gpurun_out/r4j_ab_main_$rep.err
LMPC_HIP_LIBRARY=$AB/liblmpc_fatdpp.so timeout 400 python scratch/r4_ab.py trk40 iac lmpc lmpc96 lmpc40 > gpurun_out/r4j_ab_fatdpp_$rep.jsonl 2> gpurun_out/r4j_ab_fatdpp_$rep.err
done
cat gpurun_out/r4j_ab_*.jsonl | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l)
    if 'error' in r: continue
    print('%-22s %-7s B=%-5d %-5s %.3f ms st %s it %.2f sha %s %s'%(r['lib'],r['case'],r['B'],r['prec'],r['qp_ms'],r['status'],r['iters_mean'],r['sha'],('e %.1e'%r['err_max']) if 'err_max' in r else ''))
"
