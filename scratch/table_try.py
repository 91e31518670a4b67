import sys, time; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
B = 256
def run(tag, P, s, g, mid=None):
    for _ in range(2):
        t0 = time.time(); nodes, status, iters, viol = P.plan(s, g, map_id=mid); dt = time.time() - t0
    print("  %-14s converged %3d iters hist %s wall %.2f ms" % (tag, (status == 0).sum(), np.bincount(iters), dt * 1e3))
    return nodes
cfg = PlannerConfig.knots100()
P = capi.Planner(cfg, max_batch=B)
h1, c1 = workloads.exp1_terrain(); P.set_heightfields(h1, c1)
s, g = workloads.flat_goals(B, 0)
run("flat cold", P, s, g)
for grid in ((5, 3), (9, 5)):
    dx = np.linspace(0.03, 0.15, grid[0]) * 5.0; dy = np.linspace(-0.1, 0.1, grid[1])
    P.build_init_table(dx, dy)
    x0 = P.initial_guess(s[:4], g[:4])
    nodes = run("flat table %dx%d" % grid, P, s, g)
terr = workloads.exp5_terrain(); P.set_heightfields(terr[0], terr[1])
s5, g5 = workloads.step_goals(B, seed=1, terrain=terr)
run("exp5 table", P, s5, g5)
P.set_init_table(); run("exp5 cold", P, s5, g5)
