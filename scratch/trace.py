import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
kw = {}
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = float(v) if "." in v or "e" in v else int(v)
P = capi.Planner(PlannerConfig.knots100(**kw), max_batch=64)
t = workloads.exp1_terrain(); P.set_heightfields(t[0], t[1])
s, g = workloads.flat_goals(64, 0)
r = P.plan(s, g)
print("iters", np.bincount(r[2]), "status", np.bincount(r[1]), P.timing())
for b in (0, 7, 33):
    T = P.trace(b)
    print("problem", b); print(np.array2string(np.asarray(T)[:8], precision=3, suppress_small=False))
