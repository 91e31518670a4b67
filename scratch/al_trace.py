import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
cfg = PlannerConfig.knots100()
P = capi.Planner(cfg, max_batch=256)
s, g = workloads.flat_goals(256, 0)
P.plan(s, g)
for b in (0, 1, 2, 100, 255):
    t = np.zeros((cfg.max_iter + 1, 4))
    P.lib.qtos_debug_trace(P.h, b, t.ctypes.data_as(C.POINTER(C.c_double)))
    print(b, "viol, theta, alpha, mu per iteration:", [tuple(float("%.3g" % v) for v in r) for r in t[:6]])
