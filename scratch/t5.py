import numpy as np, importlib.util, time, os, sys
spec=importlib.util.spec_from_file_location('wl','racing-lmpc-ros2_amd/workloads.py'); wl=importlib.util.module_from_spec(spec); spec.loader.exec_module(wl)
from oracle import params as P, scenario as SC, qp as Q, cbind
for name,veh,cfg,kind in (('iac40',P.iac_vehicle(),P.iac_tracking_mpc(40),'putnam'),('barc60',P.barc_vehicle(),P.barc_tracking_mpc(60),'barc'),('barc10',P.barc_vehicle(),P.barc_tracking_mpc(10),'barc')):
    tr=wl.synthetic_track(kind)
    ulo,uhi,_,_=Q.effective_bounds(cfg,veh)
    B=256
    x,u=wl.sample_initial_states(kind,B,tr['L'],ulo,uhi,1)
    inp=SC.cold_start_inputs(cfg,veh,tr,x,u,0.025)
    t0=time.time(); out=cbind.solve_batch(cfg,veh,inp); t1=time.time()
    print(name,'ms/problem %.3f'%((t1-t0)/B*1e3),'status',np.bincount(out['status']),'iters',out['iters'].min(),out['iters'].mean(),out['iters'].max(),'kkt max',out['kkt'].max(1))
    errs=[]
    for b in range(16):
        p=SC.problem(inp,b); qp=Q.build_qp(cfg,veh,p); y,info=Q.solve_dense(qp); o=qp.split(y)
        errs.append((np.abs((out['X_optm'][:,:,b]-o['X_optm'])/P.SCALE_X[:,None]).max(),np.abs((out['U_optm'][:,:,b]-o['U_optm'])/P.SCALE_U[:,None]).max(),np.abs((out['dU_optm'][:,:,b]-o['dU_optm'])/P.SCALE_U[:,None]).max(),info['status']))
    print('  max err',np.array(errs).max(0))
