import sys, os; sys.path.insert(0, '.')
os.environ["QTOS_DEBUG_LOOP"] = "1"
import numpy as np
from qtos_amd import workloads, capi
from qtos_amd.config import PlannerConfig
start, goal = workloads.flat_goals(8, seed=11)
cfg = PlannerConfig.reference_compat(duration=20.0)
P = capi.Planner(cfg, max_batch=8)
n, st, it, v = P.plan(start, goal)
print("status", st, "iters", it)
