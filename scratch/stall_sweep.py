import sys, time; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
B = 256
for stall in (5, 4, 3, 2):
    cfg = PlannerConfig.knots100(stall_iters=stall)
    P = capi.Planner(cfg, max_batch=B)
    terr = workloads.exp5_terrain(); P.set_heightfields(terr[0], terr[1])
    start, goal = workloads.step_goals(B, seed=1, terrain=terr)
    for _ in range(2):
        t0 = time.time(); nodes, status, iters, viol = P.plan(start, goal); dt = time.time() - t0
    print("exp5  stall", stall, "converged", (status == 0).sum(), "iters max", iters.max(), "mean %.2f" % iters.mean(), "wall %.1f ms" % (dt * 1e3), "non-converged viol", np.sort(viol[status != 0]).round(5))
    maps, cell = workloads.mixed_terrains(); P.set_heightfields(maps, cell)
    s, g, mid = workloads.mixed_goals(B, seed=2, terrains=(maps, cell))
    for _ in range(2):
        t0 = time.time(); nodes, status, iters, viol = P.plan(s, g, map_id=mid); dt = time.time() - t0
    print("mixed stall", stall, "converged", (status == 0).sum(), "iters max", iters.max(), "mean %.2f" % iters.mean(), "wall %.1f ms" % (dt * 1e3))
    P.close()
