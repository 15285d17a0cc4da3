import numpy as np, importlib.util, time, os, sys
spec=importlib.util.spec_from_file_location('wl','racing-lmpc-ros2_amd/workloads.py'); wl=importlib.util.module_from_spec(spec); spec.loader.exec_module(wl)
from oracle import params as P, scenario as SC, qp as Q, cbind
veh=P.barc_vehicle(); cfg=P.barc_tracking_mpc(20)
tr=wl.synthetic_track('barc')
ulo,uhi,_,_=Q.effective_bounds(cfg,veh)
B=1024
x,u=wl.sample_initial_states('barc',B,tr['L'],ulo,uhi,0)
inp=SC.cold_start_inputs(cfg,veh,tr,x,u,0.025)
out=cbind.solve_batch(cfg,veh,inp)
k=out['kkt']
print('rg big:',np.where(k[0]>1e-2)[0][:20], k[0][k[0]>1e-2][:20])
b=91
p=SC.problem(inp,b); qp=Q.build_qp(cfg,veh,p); y,info=Q.solve_dense(qp); o=qp.split(y)
yc=Q.pack(qp,out['X_optm'][:,:,b],out['U_optm'][:,:,b],out['dU_optm'][:,:,b],sigma=k[3,b])
print('obj dense',qp.objective(y),'obj C',qp.objective(yc),'sigma',o['sigma'],k[3,b])
print('cert dense',Q.kkt_certificate(qp,y)); print('cert C',Q.kkt_certificate(qp,yc))
d=(out['dU_optm'][:,:,b]-o['dU_optm'])/P.SCALE_U[:,None]; print(np.abs(d).max(0))
