#!/bin/bash
mkdir -p gpurun_out
AB=$PWD/racing-lmpc-ros2_amd/lib/ab
for lib in main fd fd2; do
  if [ $lib = main ]; then unset LMPC_HIP_LIBRARY; else export LMPC_HIP_LIBRARY=$AB/liblmpc_$lib.so; fi
  for seed in 0 1 2 3; do SEED=$seed python scratch/r4_tail32k.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['lib'],'seed $seed','max %.2e'%r['two_pass_max'],'>1e-3:',r['n_gt_1e3'],'>5e-4:',r['n_gt_5e4'],'marked',r['one_pass_marked'],'st2',r['status_two'],'st64',r['status_f64'],'lost',r['lost'])
"; done
  timeout 300 python scratch/r4_ab.py lmpc32kreg iac lmpc 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l)
    if 'err_max' in r: print(r['lib'],r['case'],r['B'],r['prec'],'reg',r['reg'],'%.3f ms'%r['qp_ms'],'max %.2e'%r['err_max'],'>5e-4:',r['n_gt_5e4'],'st',r['status'],'lost',r['lost_idx'])
"
done
