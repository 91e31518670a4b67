import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots100()
P = capi.Planner(cfg, max_batch=64)
start, goal = workloads.flat_goals(64, 0)
res = P.plan(start, goal)
np.set_printoptions(linewidth=200, precision=3)
for b in (0, 1, 7, 33):
    t = P.trace(b)
    print("problem", b, "rows: viol, theta, alpha, mu"); print(t)
