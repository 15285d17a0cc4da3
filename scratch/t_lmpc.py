import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
from pathlib import Path
from oracle import params as P, scenario as SC, qp as Q, cbind
GOLD=Path('/root/repo/tests/golden/barc_ss')
laps=[np.loadtxt(GOLD/f'ss_lap_{i}_x.txt') for i in (1,2,3)]
ks=[np.loadtxt(GOLD/f'ss_lap_{i}_k.txt') for i in (1,2,3)]
L=17.06
# track tables from lap 1
M=512; sg=np.arange(M)*L/M
order=np.argsort(laps[0][:,0]); s0=laps[0][order,0]; k0=ks[0][order]
curv=np.interp(sg, s0, k0, period=L)
tr={'L':L,'M':M,'curvature':curv,'bound_left':np.full(M,0.55),'bound_right':np.full(M,-0.55),'vel':np.full(M,2.0)}
veh=P.barc_vehicle(); cfg=P.barc_lmpc(20,3)
rng=np.random.default_rng(0)
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
idx=rng.integers(0,laps[2].shape[0],B)
x=laps[2][idx]+rng.normal(0,1,(B,6))*np.array([0.0,0.02,0.02,0.05,0.01,0.05])
x[:,0]=np.mod(x[:,0],L)
u=np.zeros((B,2))
inp=SC.cold_start_inputs(cfg,veh,tr,x,u,0.025)
# safe set query at aligned X_ref[:, -1]
from oracle import dynamics as D
q=np.stack([D.align_abscissa(inp['X_ref'][0,-1,:], inp['x_ic'][0,:], L), inp['X_ref'][1,-1,:]])
ss_x,ss_j,nf=cbind.ss_query_batch(laps,L,cfg.num_ss_pts,cfg.num_ss_pts_per_lap,q)
print('nf',np.bincount(nf))
t0=time.time(); out=cbind.solve_batch(cfg,veh,inp,ss_x=ss_x,ss_j=ss_j); t1=time.time()
print('C ms/problem %.3f'%((t1-t0)/B*1e3),'status',np.bincount(out['status'],minlength=3),'iters',np.bincount(out['iters']))
errs=[]
for b in range(min(B,12)):
    qp=Q.build_qp(cfg,veh,SC.problem(inp,b),ss_x=ss_x[:,:,b],ss_j=ss_j[:,b]); y,info=Q.solve_dense(qp); o=qp.split(y)
    errs.append((np.abs((out['X_optm'][:,:,b]-o['X_optm'])/P.SCALE_X[:,None]).max(),np.abs((out['U_optm'][:,:,b]-o['U_optm'])/P.SCALE_U[:,None]).max(),np.abs((out['dU_optm'][:,:,b]-o['dU_optm'])/P.SCALE_U[:,None]).max(),info['status'],info.get('polished',False), abs(qp.objective(y)-qp.objective(Q.pack(qp,out['X_optm'][:,:,b],out['U_optm'][:,:,b],out['dU_optm'][:,:,b],sigma=out['kkt'][3,b],lam=out['convex_combi_optm'][:,b],eps=out['X_optm'][:,-1,b]-ss_x[:,:,b]@out['convex_combi_optm'][:,b]))) ))
print(np.array(errs))
import os
os.environ['LMPC_ORACLE_DEBUG']='1'
cbind.solve_batch(cfg,veh,inp,ss_x=ss_x,ss_j=ss_j,b0=0,b1=1)
