import numpy as np, importlib.util, time, os, sys
sys.path.insert(0,'/root/repo')
from __graft_entry__ import load_package
wl=load_package().workloads
from oracle import params as P, scenario as SC, qp as Q, cbind
veh=P.barc_vehicle(); cfg=P.barc_tracking_mpc(20)
tr=wl.synthetic_track('barc')
ulo,uhi,_,_=Q.effective_bounds(cfg,veh)
B=4096
x,u=wl.sample_initial_states('barc',B,tr['L'],ulo,uhi,0)
inp=SC.cold_start_inputs(cfg,veh,tr,x,u,0.025)
out=cbind.solve_batch(cfg,veh,inp)
print(np.bincount(out['iters']))
bad=np.where(out['status']!=0)[0]; print(bad)
for b in bad:
    print('x_ic',inp['x_ic'][:,b],'u_ic',inp['u_ic'][:,b])
    qp=Q.build_qp(cfg,veh,SC.problem(inp,b)); y,info=Q.solve_dense(qp); print('dense',info['status'],info['iters'],info['mu'], Q.kkt_certificate(qp,y))
    for mi in (5,10,12,14,16,18,20,25,30):
        o=cbind.solve_batch(cfg,veh,inp,b0=b,b1=b+1,max_iter=mi)
        print(mi,o['status'][b],o['iters'][b],o['kkt'][:,b])
