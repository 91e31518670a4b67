import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
P = capi.Planner(PlannerConfig.knots100(), max_batch=256)
t = workloads.exp1_terrain(); P.set_heightfields(t[0], t[1])
s, g = workloads.flat_goals(256, 0)
for i in range(6): P.plan(s, g)
print(P.timing())
