"""Per-iteration traces (violation, slack-form violation, step length, mu) of the slowest problems of exp_5 / mixed batches."""
import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
B = 256
P = capi.Planner(PlannerConfig.knots100(), max_batch=B)
t5 = workloads.exp5_terrain(); P.set_heightfields(t5[0], t5[1])
for seed in range(1, 5):
    s, g = workloads.step_goals(B, seed=seed, terrain=t5)
    r = P.plan(s, g)
    tm = P.timing()
    print("seed", seed, "iters", np.bincount(r[2]), "kkt launches", tm["kkt_launches"], "chord", tm["chord_launches"])
    for b in np.nonzero(r[2] >= 5)[0][:6]:
        T = np.asarray(P.trace(int(b)))[:r[2][b] + 1]
        print("  problem %3d iters %d viol: %s | theta: %s | alpha: %s | mu: %s" % (b, r[2][b], " ".join("%.1e" % v for v in T[:, 0]), " ".join("%.1e" % v for v in T[:, 1]), " ".join("%.2f" % v for v in T[:, 2]), " ".join("%.0e" % v for v in T[:, 3])))
