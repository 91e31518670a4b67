import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from oracle.oracle import Oracle, oracle_dict
cfg = PlannerConfig.knots100()
O = Oracle(oracle_dict(cfg)); L = O.L
P = Planner(cfg, max_batch=4)
s, g = workloads.flat_goals(4, seed=5)
x0 = P.initial_guess(s, g)
n, st, it, v = P.plan(s, g)
tr = P.trace(0)
print(os.environ.get("QTOS_LIB"), "trace viol", tr[:, 0], "alpha", tr[:, 2])
rk, vf, order = P.structure()
zs = []
for e in range(4):
    off, cnt = L.off_eem[e], L.n_eem[e]
    seg0, segx = x0[0, off:off + cnt], n[0, off:off + cnt]
    print("foot", e, "guess", np.round(seg0, 4).tolist())
    print("foot", e, "sol  ", np.round(segx, 4).tolist())
    print("foot", e, "free ", vf[off:off + cnt].tolist())
