import sys, time; sys.path.insert(0, '.')
import numpy as np
from oracle.oracle import Oracle
from qtos_amd import workloads
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots200()
start, goal, mid = workloads.mpc_goals(8)
O = Oracle(cfg.oracle_dict())
print("n", O.n, "m", O.m)
for b in range(2):
    s, g = start[b], goal[b]
    t0 = time.time(); x, info = O.solve(O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g, (0, 0, 0), (0, 0, 0), 0.0)); print("flat", info.status, info.iters, info.inf_pr, "%.2f s" % (time.time() - t0))
maps, cell = workloads.random_terrains()
print("terrain range", maps.min(), maps.max())
for b in range(2):
    O = Oracle(cfg.oracle_dict(), height=maps[mid[b]], hcell=cell)
    s, g = start[b], goal[b]
    t0 = time.time(); x, info = O.solve(O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g, (0, 0, 0), (0, 0, 0), 0.0)); print("random", info.status, info.iters, info.inf_pr, "%.2f s" % (time.time() - t0))
