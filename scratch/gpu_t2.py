import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from oracle.oracle import Oracle
cfg = PlannerConfig.reference_compat()
hxy, cell = workloads.exp5_terrain()
start, goal = workloads.step_goals(64, seed=1, terrain=(hxy, cell))
P = Planner(cfg, max_batch=64); P.set_heightfields(hxy, cell)
nodes, status, iters, viol = P.plan(start, goal)
print('status', np.bincount(status), 'iters', iters)
O = Oracle(cfg.oracle_dict(), height=hxy, hcell=cell)
for b in range(6):
    s, g = start[b], goal[b]
    q = O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g)
    o = O.default_options(); o.verbose = 1 if b == 2 else 0
    xo, info = O.solve(q, opts=o)
    print(b, 'gpu', status[b], iters[b], 'cpu', info.status, info.iters, 'diff', np.abs(nodes[b]-xo).max())
    if b == 2:
        print(P.trace(b))
