import sys; sys.path.insert(0, '.')
import numpy as np
from qtos_amd import workloads
from qtos_amd.capi import Planner
from qtos_amd.config import PlannerConfig
s, g = workloads.flat_goals(256, 0)
terr5 = workloads.exp5_terrain(); s5, g5 = workloads.step_goals(256, seed=1, terrain=terr5)
mt = workloads.mixed_terrains(); sm, gm, mm = workloads.mixed_goals(256, seed=2, terrains=mt)
for kw in (dict(), dict(delta_x=1e-3), dict(delta_x=1e-4), dict(mu_init=0.01), dict(delta_x=1e-3, mu_init=0.01), dict(slack_push=0.3), dict(slack_push=0.1), dict(delta_x=3e-3)):
    cfg = PlannerConfig.knots100(**kw)
    P = Planner(cfg, max_batch=256)
    hxy, cell = workloads.exp1_terrain(); P.set_heightfields(hxy, cell)
    n, st, it, v = P.plan(s, g)
    tr = P.trace(0)
    P.set_heightfields(terr5[0], terr5[1]); n5, st5, it5, v5 = P.plan(s5, g5)
    P.set_heightfields(mt[0], mt[1]); nm, stm, itm, vm = P.plan(sm, gm, mm)
    print(kw, "flat: conv %d iters %s | exp5: conv %d max %d mean %.2f | mixed: conv %d max %d mean %.2f" % ((st == 0).sum(), np.bincount(it), (st5 == 0).sum(), it5.max(), it5.mean(), (stm == 0).sum(), itm.max(), itm.mean()))
    print("   trace viol:", " ".join("%.2e" % t for t in tr[:, 0]), "| alpha:", " ".join("%.2f" % t for t in tr[:, 2]), "| mu:", " ".join("%.1e" % t for t in tr[:, 3]))
    P.close()
