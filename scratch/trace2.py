import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
kw = {}
for a in sys.argv[2:]:
    k, v = a.split("="); kw[k] = float(v) if "." in v or "e" in v else int(v)
which = sys.argv[1]
B = 256
if which == "knots200":
    P = capi.Planner(PlannerConfig.knots200(**kw), max_batch=B); t = workloads.exp1_terrain(); P.set_heightfields(t[0], t[1]); s, g = workloads.flat_goals(B, 0); m = None
elif which == "exp5":
    P = capi.Planner(PlannerConfig.knots100(**kw), max_batch=B); t = workloads.exp5_terrain(); P.set_heightfields(t[0], t[1]); s, g = workloads.step_goals(B, seed=1, terrain=t); m = None
else:
    P = capi.Planner(PlannerConfig.knots100(**kw), max_batch=B); t = workloads.mixed_terrains(); P.set_heightfields(t[0], t[1]); s, g, m = workloads.mixed_goals(B, seed=2, terrains=t)
r = P.plan(s, g, map_id=m)
print(which, kw, "iters", np.bincount(r[2]), "status", np.bincount(r[1]), {k: (round(v, 5) if isinstance(v, float) else v) for k, v in P.timing().items()})
order = np.argsort(-r[2])
for b in list(order[:4]) + list(order[-1:]):
    T = np.asarray(P.trace(int(b)))[:r[2][b] + 1]
    print(" problem", b, "iters", r[2][b], "viol:", " ".join("%.1e" % v for v in T[:, 0]), "| alpha:", " ".join("%.2f" % v for v in T[:, 2]))
