import os, sys
sys.path.insert(0, os.path.dirname(__file__)); 
from common import *
import dense_cases as DC
name, b = sys.argv[1], int(sys.argv[2])
cfg, veh, inp, ss_x, ss_j = DC.build(pkg, name)
kw = {} if ss_x is None else {"ss_x": ss_x[:, :, b], "ss_j": ss_j[:, b]}
qp = Q.build_qp(cfg, veh, S.problem(inp, b), **kw)
y, info = Q.solve_dense(qp)
print({k: v for k, v in info.items() if k not in ("lam", "pi")})
