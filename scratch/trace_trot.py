import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
kw = {}
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = float(v) if "." in v or "e" in v else int(v)
P = capi.Planner(PlannerConfig.knots100(gait="trot", **kw), max_batch=256)
t = workloads.exp1_terrain(); P.set_heightfields(t[0], t[1])
s, g = workloads.flat_goals(256, 0)
r = P.plan(s, g)
print(kw, "iters", np.bincount(r[2]), "status", np.bincount(r[1]), P.timing())
for b in list(np.nonzero(r[2] == 4)[0][:2]) + list(np.nonzero(r[2] == 5)[0][:3]):
    T = np.asarray(P.trace(int(b)))[:7]
    print("problem", b, "iters", r[2][b]); print(np.array2string(T, precision=3))
