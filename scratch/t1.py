import numpy as np, importlib.util, time, os, sys
spec=importlib.util.spec_from_file_location('wl','racing-lmpc-ros2_amd/workloads.py'); wl=importlib.util.module_from_spec(spec); spec.loader.exec_module(wl)
from oracle import params as P, scenario as SC, qp as Q, cbind
veh=P.barc_vehicle(); cfg=P.barc_tracking_mpc(20)
tr=wl.synthetic_track('barc')
ulo,uhi,_,_=Q.effective_bounds(cfg,veh)
B=int(sys.argv[1]) if len(sys.argv)>1 else 1024
nd=int(sys.argv[2]) if len(sys.argv)>2 else 64
x,u=wl.sample_initial_states('barc',B,tr['L'],ulo,uhi,0)
inp=SC.cold_start_inputs(cfg,veh,tr,x,u,0.025)
t0=time.time(); out=cbind.solve_batch(cfg,veh,inp); t1=time.time()
print('C solve time per problem %.3f ms'%((t1-t0)/B*1e3))
print('status',np.bincount(out['status']),'iters',out['iters'].min(),out['iters'].mean(),out['iters'].max(), np.bincount(out['iters']))
print('kkt max',out['kkt'].max(1))
bad=np.where(out['status']!=0)[0]; print('bad',bad[:10])
errs=[]
for b in range(nd):
    p=SC.problem(inp,b); qp=Q.build_qp(cfg,veh,p); y,info=Q.solve_dense(qp); o=qp.split(y)
    ex=np.abs((out['X_optm'][:,:,b]-o['X_optm'])/P.SCALE_X[:,None]).max()
    eu=np.abs((out['U_optm'][:,:,b]-o['U_optm'])/P.SCALE_U[:,None]).max()
    ed=np.abs((out['dU_optm'][:,:,b]-o['dU_optm'])/P.SCALE_U[:,None]).max()
    errs.append((ex,eu,ed, abs(out['kkt'][3,b]-o['sigma'])))
errs=np.array(errs); print('max err',errs.max(0)); print('worst',np.argsort(-errs.max(1))[:5])
print('err percentiles (max over X,U,dU scaled):')
e=errs[:,:3].max(1)
for q in (50,75,90,95,99,100): print(q, '%.2e'%np.percentile(e,q))
pol=out['kkt'][2,:nd]==0

print('polished overall', (out['kkt'][2]==0).mean())
