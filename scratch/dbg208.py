import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import workloads, capi
from qtos_amd.config import PlannerConfig
start, goal = workloads.flat_goals(8, seed=11)
for kw in (dict(duration=20.0), dict(duration=20.0, chord_tol=0.0), dict(duration=12.0), dict(duration=10.0, dt_base=0.05, dt_dynamic=0.05)):
    cfg = PlannerConfig.reference_compat(**kw)
    P = capi.Planner(cfg, max_batch=8)
    n, st, it, v = P.plan(start, goal)
    print(kw, "status", st, "iters", it, "viol", np.array2string(v, precision=1))
    for b in np.nonzero(st != 0)[0][:1]:
        tr = P.trace(int(b)); print("  trace viol", " ".join("%.2e" % t for t in tr[:, 0]), "alpha", " ".join("%.2f" % t for t in tr[:, 2]))
    P.close()
