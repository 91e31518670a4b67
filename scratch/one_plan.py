import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dataclasses
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
cfg = dataclasses.replace(PlannerConfig.knots100(), chord_tol=0.0)
P = capi.Planner(cfg, max_batch=256)
s, g = workloads.flat_goals(256, 0)
for i in range(3):
    P.plan(s, g)
t = P.timing()
print("kkt ms/launch %.4f launches %d" % (1e3 * t["kkt_seconds"] / t["kkt_launches"], t["kkt_launches"]))
