import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
for gait in ("walk", "trot"):
    for sl in (True, False):
        P = capi.Planner(PlannerConfig.knots100(gait=gait, mu_superlinear=sl), max_batch=256)
        s, g = workloads.flat_goals(256, 0)
        r = P.plan(s, g)
        V = np.array([np.asarray(P.trace(b))[:5, 0] for b in range(256)])
        A = np.array([np.asarray(P.trace(b))[:5, 2] for b in range(256)])
        print(gait, "superlinear" if sl else "plain", "iters", np.bincount(r[2]))
        for it in range(V.shape[1]):
            print("   iterate %d: viol min %.2e median %.2e max %.2e | alpha min %.2f median %.2f" % (it, V[:, it].min(), np.median(V[:, it]), V[:, it].max(), A[:, it].min(), np.median(A[:, it])))
        P.close()
