import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
from pathlib import Path
from oracle import params as P, scenario as SC, qp as Q, cbind, dynamics as D
GOLD=Path('/root/repo/tests/golden/barc_ss')
laps=[np.loadtxt(GOLD/f'ss_lap_{i}_x.txt') for i in (1,2,3)]
ks=[np.loadtxt(GOLD/f'ss_lap_{i}_k.txt') for i in (1,2,3)]
L=17.06; M=512; sg=np.arange(M)*L/M
order=np.argsort(laps[0][:,0]); curv=np.interp(sg, laps[0][order,0], ks[0][order], period=L)
tr={'L':L,'M':M,'curvature':curv,'bound_left':np.full(M,0.55),'bound_right':np.full(M,-0.55),'vel':np.full(M,2.0)}
veh=P.barc_vehicle(); cfg=P.barc_lmpc(20,3)
rng=np.random.default_rng(1); B=256
idx=rng.integers(0,laps[2].shape[0],B)
x=laps[2][idx]+rng.normal(0,1,(B,6))*np.array([0.0,0.03,0.03,0.1,0.02,0.1]); x[:,0]=np.mod(x[:,0],L)
inp=SC.cold_start_inputs(cfg,veh,tr,x,np.zeros((B,2)),0.025)
q=np.stack([D.align_abscissa(inp['X_ref'][0,-1,:], inp['x_ic'][0,:], L), inp['X_ref'][1,-1,:]])
ss_x,ss_j,nf=cbind.ss_query_batch(laps,L,cfg.num_ss_pts,cfg.num_ss_pts_per_lap,q)
t0=time.time(); out=cbind.solve_batch(cfg,veh,inp,ss_x=ss_x,ss_j=ss_j); t1=time.time()
print('C ms/problem %.3f'%((t1-t0)/B*1e3),'status',np.bincount(out['status'],minlength=3),'iters mean %.2f max %d'%(out['iters'].mean(),out['iters'].max()))
errs=[]
for b in range(24):
    qp=Q.build_qp(cfg,veh,SC.problem(inp,b),ss_x=ss_x[:,:,b],ss_j=ss_j[:,b]); y,info=Q.solve_dense(qp); o=qp.split(y)
    errs.append((np.abs((out['X_optm'][:,:,b]-o['X_optm'])/P.SCALE_X[:,None]).max(),np.abs((out['U_optm'][:,:,b]-o['U_optm'])/P.SCALE_U[:,None]).max(),np.abs((out['dU_optm'][:,:,b]-o['dU_optm'])/P.SCALE_U[:,None]).max(),info['status']))
e=np.array(errs); print('max err X,U,dU',e.max(0)[:3],'median',np.median(e,0)[:3],'dense fail',int(e[:,3].sum()))
