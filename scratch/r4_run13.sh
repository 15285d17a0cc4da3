#!/bin/bash
mkdir -p gpurun_out
AB=$PWD/racing-lmpc-ros2_amd/lib/ab
timeout 300 python scratch/r4_ab.py trk20 trk40 trk60 lmpc lmpc32kreg iac > gpurun_out/r4m_ab_main.jsonl 2> gpurun_out/r4m_ab_main.err
for st in max-ilp max-memory-clause iterative-minreg; do
LMPC_HIP_LIBRARY=$AB/liblmpc_sched_$st.so timeout 300 python scratch/r4_ab.py trk20 trk40 trk60 lmpc lmpc32kreg iac > gpurun_out/r4m_ab_$st.jsonl 2> gpurun_out/r4m_ab_$st.err
done
timeout 300 python scratch/r4_ab.py trk20 trk40 trk60 lmpc lmpc32kreg iac > gpurun_out/r4m_ab_main2.jsonl 2> gpurun_out/r4m_ab_main2.err
cat gpurun_out/r4m_ab_*.jsonl | python -c "
import sys,json,collections
d=collections.OrderedDict()
for l in sys.stdin:
    r=json.loads(l)
    if 'error' in r: continue
    d.setdefault((r['case'],r['B'],r['prec']),[]).append((r['lib'],r['qp_ms'],r['sha'],r['status']))
for k,v in d.items():
    print(k, ' | '.join('%s %.3f %s'%(a.replace('liblmpc_','').replace('.so',''),b,c[:6]) for a,b,c,s in v))
"
