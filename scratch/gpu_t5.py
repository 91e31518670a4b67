import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from oracle.oracle import Oracle
cfg = PlannerConfig.reference_compat()
maps, cell = workloads.mixed_terrains()
start, goal, map_id = workloads.mixed_goals(96, seed=2, terrains=(maps, cell))
P = Planner(cfg, max_batch=96); P.set_heightfields(maps, cell)
nodes, status, iters, viol = P.plan(start, goal, map_id=map_id)
print(np.bincount(status), 'by map', [np.bincount(status[map_id==k], minlength=3).tolist() for k in range(3)])
for b in np.nonzero(status == 2)[0]:
    O = Oracle(cfg.oracle_dict(), height=maps[map_id[b]], hcell=cell)
    s, g = start[b], goal[b]
    x, info = O.solve(O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g))
    tr = P.trace(int(b))
    print(b, 'map', map_id[b], 'gpu iters', iters[b], 'viol', viol[b], 'oracle', info.status, info.iters, info.inf_pr)
    print(tr[-3:])
