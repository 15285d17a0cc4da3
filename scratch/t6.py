import numpy as np, importlib.util, time, os, sys
spec=importlib.util.spec_from_file_location('wl','racing-lmpc-ros2_amd/workloads.py'); wl=importlib.util.module_from_spec(spec); spec.loader.exec_module(wl)
from oracle import params as P, scenario as SC, qp as Q, cbind
veh=P.barc_vehicle(); cfg=P.barc_tracking_mpc(60); kind='barc'
tr=wl.synthetic_track(kind)
ulo,uhi,_,_=Q.effective_bounds(cfg,veh)
B=256
x,u=wl.sample_initial_states(kind,B,tr['L'],ulo,uhi,1)
inp=SC.cold_start_inputs(cfg,veh,tr,x,u,0.025)
out=cbind.solve_batch(cfg,veh,inp)
bad=np.where(out['status']!=0)[0]; print(bad)
b=bad[0]; print('x_ic',inp['x_ic'][:,b]); print('Xref vx',inp['X_ref'][3,:,b][::6]); print('Xref ey',inp['X_ref'][1,:,b][::6]); print('xref om', inp['X_ref'][5,:,b][::6])
A,Bm,g=cbind.linearize_batch(cfg,veh,inp)
Ab=A[:,:,:,b].transpose(2,0,1)
print('max |eig A| per stage', [round(abs(np.linalg.eigvals(Ab[i])).max(),2) for i in range(0,59,6)])
for mi in (0,1,2,3,5,8,12,20,40):
    o=cbind.solve_batch(cfg,veh,inp,b0=b,b1=b+1,max_iter=mi if mi else 1)
    print(mi,o['status'][b],o['iters'][b],o['kkt'][:,b], np.abs(o['X_optm'][:,:,b]).max())
