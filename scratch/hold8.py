import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
for hold in (0, 2):
    cfg = PlannerConfig.reference_compat(duration=8.0, foothold_hold_from=hold)
    P = capi.Planner(cfg, max_batch=8)
    start, goal = workloads.flat_goals(8, seed=11)
    goal[:, 0] = start[:, 0] + (goal[:, 0] - start[:, 0]) * cfg.duration / 5.0
    nodes, status, iters, viol = P.plan(start, goal)
    print("hold", hold, "status", status, "iters", iters, "viol", viol.round(6))
    t = P.trace(0)
    print("   problem 0 viol:", " ".join("%.1e" % v for v in t[:iters[0] + 1, 0]), "| alpha:", " ".join("%.2f" % v for v in t[:iters[0] + 1, 2]))
    P.close()
