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
bad=np.where(out['status']!=0)[0]; print(bad[:10], out['iters'][bad[:10]])
os.environ['LMPC_ORACLE_DEBUG']='1'
b=bad[0]
cbind.solve_batch(cfg,veh,inp,b0=b,b1=b+1)
