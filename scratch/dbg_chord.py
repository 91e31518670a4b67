import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from qtos_amd import workloads, capi
from qtos_amd.config import PlannerConfig
rng = np.random.default_rng(1)
for kw in (dict(), dict(duration=12.0), dict(duration=20.0), dict(duration=2.5)):
    cfg = PlannerConfig.reference_compat(**kw)
    P = capi.Planner(cfg, max_batch=4)
    start, goal = workloads.flat_goals(4, seed=11)
    x = P.initial_guess(start, goal) + 0.01 * rng.standard_normal((4, P.n))
    rk, vf, _ = P.structure()
    I = rk == 2
    sig = np.zeros((4, P.m)); w = np.zeros((4, P.m))
    sig[:, I] = 10.0 ** rng.uniform(-3, 3, (4, I.sum())); w[:, I] = rng.standard_normal((4, I.sum()))
    dx = P.debug_newton(start, goal, x, sig, w)
    dc = P.debug_chord(4)
    print(kw, "front", P.dims.front, "max |dx|", np.abs(dx).max(), "max |chord - full|", np.abs(dc - dx).max(), "rel", np.abs(dc - dx).max() / np.abs(dx).max())
    P.close()
