import sys, time, os; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
B = 256
def run(tag, P, start, goal, mid=None):
    for _ in range(2):
        t0 = time.time(); nodes, status, iters, viol = P.plan(start, goal, map_id=mid); dt = time.time() - t0
    print("  %-6s converged %3d iters max %2d mean %.2f wall %.1f ms" % (tag, (status == 0).sum(), iters.max(), iters.mean(), dt * 1e3), "hist", np.bincount(iters)[4:])
print("FROM", os.environ.get("QTOS_FREEZE_FROM"), "weight", os.environ.get("QTOS_FREEZE_WEIGHT"), "mode", os.environ.get("QTOS_FREEZE_MODE"))
cfg = PlannerConfig.knots100()
P = capi.Planner(cfg, max_batch=B)
terr = workloads.exp5_terrain(); P.set_heightfields(terr[0], terr[1])
s, g = workloads.step_goals(B, seed=1, terrain=terr); run("exp5", P, s, g)
maps, cell = workloads.mixed_terrains(); P.set_heightfields(maps, cell)
s, g, mid = workloads.mixed_goals(B, seed=2, terrains=(maps, cell)); run("mixed", P, s, g, mid)
h1, c1 = workloads.exp1_terrain(); P.set_heightfields(h1, c1)
s, g = workloads.flat_goals(B, 0); run("flat", P, s, g)
P.close()
cfg = PlannerConfig.knots200()
P = capi.Planner(cfg, max_batch=B)
maps, cell = workloads.random_terrains(); P.set_heightfields(maps, cell)
s, g, mid = workloads.mpc_goals(B, terrains=(maps, cell)); run("mpc200", P, s, g, mid)
