import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots100()
B = 256
P = capi.Planner(cfg, max_batch=B)
terr = workloads.exp5_terrain(); P.set_heightfields(terr[0], terr[1])
start, goal = workloads.step_goals(B, seed=1, terrain=terr)
nodes, status, iters, viol = P.plan(start, goal)
print("status", np.bincount(status, minlength=3), "iters hist", np.bincount(iters))
bad = np.nonzero((status != 0) | (iters > 7))[0]
for b in bad[:6]:
    t = P.trace(b)
    print("problem", b, "status", status[b], "iters", iters[b])
    for i in range(iters[b] + 1):
        print("   it %2d viol %.3e theta %.3e alpha %.3f mu %.1e" % (i, t[i, 0], t[i, 1], t[i, 2], t[i, 3]))
