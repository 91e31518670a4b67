import sys, time; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots200()
NB = 256
P = capi.Planner(cfg, max_batch=NB)
print("dims", P.dims.n_vars, P.dims.n_stages, P.dims.front)
terr0 = workloads.random_terrains()
start, goal, mid = workloads.mpc_goals(NB, terrains=terr0)
goal_flat = goal.copy()
for tag, terr in (("random", terr0),):
    if terr is not None:
        P.set_heightfields(terr[0], terr[1])
    for rep in range(2):
        t0 = time.time(); r = P.plan(start, goal, map_id=mid if terr is not None else None); dt = time.time() - t0
    nodes, status, iters, viol = r[:4]
    print(tag, "status", np.bincount(status, minlength=3), "iters", np.bincount(iters), "viol max %.2e" % viol.max(), "wall %.1f ms" % (dt * 1e3), P.timing())
    # warm restart from the solution (fixed point)
    r2 = P.plan(start, goal, map_id=mid if terr is not None else None, warm=nodes)
    print("  warm: iters", np.bincount(r2[2]), "status", np.bincount(r2[1], minlength=3), P.timing())
