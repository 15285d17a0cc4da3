#!/bin/bash
# round 4, GPU call 7: the lean vector solve with DPP broadcasts against the LDS exchange, long horizons, checksums, repeats
mkdir -p gpurun_out
AB=$PWD/racing-lmpc-ros2_amd/lib/ab
for rep in 1 2; do
timeout 400 python scratch/r4_ab.py trk60 trk80 lmpc60 lmpc80 iac80 > gpurun_out/r4g_ab_main_$rep.jsonl 2> gpurun_out/r4g_ab_main_$rep.err
LMPC_HIP_LIBRARY=$AB/liblmpc_leandpp.so timeout 400 python scratch/r4_ab.py trk60 trk80 lmpc60 lmpc80 iac80 > gpurun_out/r4g_ab_leandpp_$rep.jsonl 2> gpurun_out/r4g_ab_leandpp_$rep.err
done
cat gpurun_out/r4g_ab_*.jsonl | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l)
    if 'error' in r: print(r['lib'],r['case'],r['error'][:80]); continue
    print('%-22s %-7s %-5s %.3f ms st %s it %.2f sha %s %s'%(r['lib'],r['case'],r['prec'],r['qp_ms'],r['status'],r['iters_mean'],r['sha'],('e %.1e'%r['err_max']) if 'err_max' in r else ''))
"
