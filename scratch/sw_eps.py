"""Where the GPU <-> oracle gap of the reduced swings comes from: the oracle keeps the swing rows with -eps_dual on their
multipliers; the gap as a function of the ORACLE's eps_dual (the product untouched)."""
import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
from oracle.oracle import Oracle, oracle_dict, oracle_options
cases = [('walk', PlannerConfig.knots100(), None), ('duration8', PlannerConfig.reference_compat(duration=8.0), None), ('knots200 random', PlannerConfig.knots200(), 'rand')]
for name, cfg, ter in cases:
    B = 4
    P = Planner(cfg, max_batch=B)
    if ter:
        maps, cell = workloads.random_terrains()
        P.set_heightfields(maps, cell)
        start, goal, mid = workloads.mpc_goals(B, seed=5, terrains=(maps, cell))
    else:
        start, goal = workloads.flat_goals(B, seed=0); mid = None
    nodes, status, iters, viol = P.plan(start, goal, map_id=mid)
    for eps in (1e-8, 1e-9, 1e-10):
        worst = 0
        for b in range(B):
            O = Oracle(oracle_dict(cfg), height=None if not ter else maps[mid[b]], hcell=0.1 if not ter else cell)
            o = oracle_options(cfg, O); o.eps_dual = eps
            x, info = O.solve(O.problem(start[b][0:3], start[b][3:6], start[b][6:18].reshape(4, 3), goal[b]), opts=o)
            if info.iters == iters[b]:
                worst = max(worst, np.abs(x - nodes[b]).max())
        print(name, 'oracle eps_dual %.0e' % eps, 'max |gpu - oracle| %.2e' % worst, 'iters', iters.tolist())
    P.close()
