import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
B = 256
P = capi.Planner(PlannerConfig.knots200(chord_tol=0.0), max_batch=B)
t = workloads.random_terrains(); P.set_heightfields(t[0], t[1])
for seed in range(3):
    s, g, m = workloads.mpc_goals(B, seed=5 + 1000 * seed, terrains=t)
    r = P.plan(s, g, map_id=m)
    print("seed", seed, "iters", np.bincount(r[2]), "status", np.bincount(r[1]))
    for b in np.nonzero(r[2] >= 7)[0][:4]:
        T = np.asarray(P.trace(int(b)))[:r[2][b] + 1]
        print(" problem", b, "status", r[1][b], "iters", r[2][b], "viol:", " ".join("%.1e" % v for v in T[:, 0]), "| alpha:", " ".join("%.2f" % v for v in T[:, 2]))
