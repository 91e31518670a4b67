"""diagnostic: GPU vs oracle node differences of the `-duration` cases of test_other_horizons_match_oracle for the library QTOS_LIB names"""
import sys; sys.path.insert(0, '.')
import numpy as np
from oracle.oracle import Oracle, oracle_dict
from qtos_amd import capi, workloads
from qtos_amd.config import PlannerConfig
for kw in (dict(duration=8.0), dict(duration=12.0), dict(duration=20.0), dict(duration=2.5), dict()):
    cfg = PlannerConfig.reference_compat(**kw)
    P = capi.Planner(cfg, max_batch=8)
    O = Oracle(oracle_dict(cfg))
    start, goal = workloads.flat_goals(8, seed=11)
    goal[:, 0] = start[:, 0] + (goal[:, 0] - start[:, 0]) * (cfg.duration / 5.0 if cfg.duration <= 8.0 else 1.0)
    nodes, status, iters, viol = P.plan(start, goal)
    errs = []
    for b in range(4):
        s, g = start[b], goal[b]
        xo, info = O.solve(O.problem(s[0:3], s[3:6], s[6:18].reshape(4, 3), g, (0, 0, 0), (0, 0, 0), 0.0))
        d = np.abs(nodes[b] - xo)
        errs.append((float(d.max()), float((d / np.maximum(1.0, np.abs(xo))).max()), int(iters[b]), info.iters))
    print(kw, ["abs %.2e rel %.2e it %d/%d" % e for e in errs])
    P.close()
